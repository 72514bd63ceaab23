"""Where the wave cycles of each kernel go, from ONE rocprofv3 --pmc pass over the SQ block:
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES \\
            --kernel-trace --output-format csv -d DIR -o p -- <command>
  python tools/pmc_sq_summary.py DIR ["header line"]
Per kernel (mean per launch): share of the wave cycles parked on s_waitcnt / barriers (WAIT_ANY), stalled at issue
(WAIT_INST_ANY), issuing (ACTIVE_INST_ANY; of which VALU / LDS), VALU instructions per wave.  MI355X_MICROARCH.md "rocprofv3 PMC
slots": WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES (disjoint), all in quad-cycles."""
import csv
import glob
import os
import re
import sys


def short(name):
  name = re.sub(r'\(anonymous namespace\)::', '', name)
  name = re.sub(r'^void ', '', name)
  return re.sub(r'\(gnpde::.*|\(float.*|\(int.*|\(.*', '', name)


def main():
  d = sys.argv[1]
  acc = {}
  for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(path)):
      k = short(r['Kernel_Name'])
      if 'gnpde::' not in r['Kernel_Name']:
        continue
      ent = acc.setdefault(k, {}).setdefault(r['Counter_Name'], [0, 0.0])
      ent[0] += 1
      ent[1] += float(r['Counter_Value'])
  if len(sys.argv) > 2:
    print('# ' + sys.argv[2])
  print('# kernel, launches, wave quad-cycles per launch, parked (s_waitcnt/barrier) %, issue-stalled %, issuing % (VALU % / LDS %), VALU instructions per wave')
  rows = []
  for k, cs in acc.items():
    m = {c: v[1] / v[0] for c, v in cs.items()}
    wc = m.get('SQ_WAVE_CYCLES', 0.0)
    if wc <= 0:
      continue
    n = max(v[0] for v in cs.values())
    pct = lambda c: 100.0 * m.get(c, 0.0) / wc   # noqa: E731
    rows.append((wc * n, '%-72s %5d %12.0f   parked %5.1f   stalled %5.1f   issuing %5.1f (VALU %5.1f / LDS %5.1f)   VALU/wave %7.1f' % (
      k[:72], n, wc, pct('SQ_WAIT_ANY'), pct('SQ_WAIT_INST_ANY'), pct('SQ_ACTIVE_INST_ANY'), pct('SQ_ACTIVE_INST_VALU'), pct('SQ_ACTIVE_INST_LDS'),
      m.get('SQ_INSTS_VALU', 0.0) / max(m.get('SQ_WAVES', 1.0), 1.0))))
  for _, line in sorted(rows, reverse=True):
    print(line)


if __name__ == '__main__':
  main()
