"""Register / LDS / scratch budget and compiler-reported occupancy of every kernel of csrc/*.hip (no GPU needed).

  python tools/kernel_resources.py [OUT.txt]        (default: profiles/kernel_resources.txt)

Recompiles each .hip file for gfx950 with -Rpass-analysis=kernel-resource-usage (objects go to a temporary directory, the
library is untouched) and prints one line per kernel instantiation: VGPRs, AGPRs, SGPRs, scratch bytes per lane, spills, LDS
bytes per workgroup and the occupancy the compiler derives from them (waves per SIMD; 8 is the gfx950 maximum at <= 64 VGPRs).
Kernels that spill or use scratch are listed again at the end -- a hot kernel there is the first thing to fix.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'graph-neural-pde_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FIELDS = [('VGPRs', 'vgpr'), ('AGPRs', 'agpr'), ('TotalSGPRs', 'sgpr'), ('ScratchSize [bytes/lane]', 'scratch'),
          ('Occupancy [waves/SIMD]', 'occ'), ('SGPRs Spill', 'sspill'), ('VGPRs Spill', 'vspill'),
          ('LDS Size [bytes/block]', 'lds')]


def demangle(names):
  if not names:
    return {}
  for filt in ('/opt/rocm/lib/llvm/bin/llvm-cxxfilt', 'c++filt'):
    try:
      out = subprocess.run([filt], input='\n'.join(names) + '\n', capture_output=True, text=True, check=True).stdout.split('\n')
      return {n: o for n, o in zip(names, out)}
    except Exception:
      continue
  return {n: n for n in names}


def short(name):
  name = re.sub(r'\(anonymous namespace\)::', '', name)
  name = re.sub(r'^void ', '', name)
  return name.split('(')[0]


def analyse(path, tmp):
  cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
         '-Wno-unused-function', '-Wno-pass-failed', '-Rpass-analysis=kernel-resource-usage', '-c', path, '-o',
         os.path.join(tmp, os.path.basename(path) + '.o')]
  err = subprocess.run(cmd, capture_output=True, text=True).stderr
  recs, cur = [], None
  for line in err.split('\n'):
    m = re.search(r'remark: Function Name: (\S+)', line)
    if m:
      cur = {'mangled': m.group(1)}
      recs.append(cur)
      continue
    if cur is None:
      continue
    for label, key in FIELDS:
      m = re.search(r'remark:\s+' + re.escape(label) + r': (\d+)', line)
      if m:
        cur[key] = int(m.group(1))
  names = demangle([r['mangled'] for r in recs])
  for r in recs:
    r['name'] = short(names.get(r['mangled'], r['mangled']))
    r['file'] = os.path.basename(path)
  return recs


def main():
  out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'kernel_resources.txt')
  files = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
  rows = []
  with tempfile.TemporaryDirectory() as tmp:
    for f in files:
      rows += analyse(os.path.join(CSRC, f), tmp)
  rows = [r for r in rows if 'vgpr' in r]
  lines = ['# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage over graph-neural-pde_amd/csrc/*.hip',
           '# (tools/kernel_resources.py; occupancy = waves per SIMD as the compiler derives it from VGPRs + AGPRs and LDS)',
           '%-12s %-86s %5s %5s %5s %7s %6s %7s %4s' % ('file', 'kernel', 'VGPR', 'AGPR', 'SGPR', 'scratch', 'spills', 'LDS', 'occ')]
  for r in sorted(rows, key=lambda r: (r['file'], r['name'])):
    lines.append('%-12s %-86s %5d %5d %5d %7d %6d %7d %4d' % (
      r['file'], r['name'][:86], r['vgpr'], r.get('agpr', 0), r.get('sgpr', 0), r.get('scratch', 0),
      r.get('sspill', 0) + r.get('vspill', 0), r.get('lds', 0), r.get('occ', 0)))
  bad = [r for r in rows if r.get('scratch', 0) or r.get('vspill', 0)]
  lines.append('')
  lines.append('# kernels with scratch memory or VGPR spills: %d of %d' % (len(bad), len(rows)))
  for r in sorted(bad, key=lambda r: (r['file'], r['name'])):
    lines.append('#   %s %s: scratch %d B/lane, VGPR spills %d' % (r['file'], r['name'][:100], r.get('scratch', 0), r.get('vspill', 0)))
  text = '\n'.join(lines) + '\n'
  open(out_path, 'w').write(text)
  print(text if len(lines) < 60 else '\n'.join(lines[:3] + ['...'] + lines[-(len(bad) + 3):]))
  print('%d kernel instantiations -> %s' % (len(rows), out_path))


if __name__ == '__main__':
  main()
