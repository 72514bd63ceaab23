// Thread-local error text + ABI version.
#include "common.h"

namespace gnpde {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gnpde

namespace gnpde { int g_tune[GNPDE_TUNE_COUNT] = {0}; }

extern "C" int gnpde_tune(int32_t key, int32_t value) {
  GNPDE_CHECK_ARG(key >= 0 && key < gnpde::GNPDE_TUNE_COUNT, GNPDE_EINVAL, "tune: bad key %d", key);
  gnpde::g_tune[key] = value;
  return 0;
}

extern "C" int gnpde_abi_version(void) { return GNPDE_ABI_VERSION; }
extern "C" const char* gnpde_last_error(void) { return gnpde::g_err; }
