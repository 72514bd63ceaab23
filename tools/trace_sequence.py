"""Print one window of a rocprofv3 kernel trace in launch order: kernel (short name), duration, gap to the previous kernel's end.
usage: trace_sequence.py <kernel_trace.csv> <anchor substring> [occurrence] [count]
The window starts at the `occurrence`-th kernel whose name contains the anchor (default: the middle one)."""
import csv
import re
import sys

path, anchor = sys.argv[1], sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else -1
count = int(sys.argv[4]) if len(sys.argv) > 4 else 40
rows = []
with open(path) as f:
  for r in csv.DictReader(f):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
hits = [i for i, r in enumerate(rows) if anchor in r[2]]
if not hits:
  raise SystemExit('no kernel matches %r' % anchor)
start = hits[len(hits) // 2] if occ < 0 else hits[occ]
prev_end = rows[start - 1][1] if start > 0 else rows[start][0]
total = 0
for s, e, name in rows[start:start + count]:
  short = re.sub(r'\(gnpde::.*|\(float.*|\(int.*', '', name.replace('void ', '').replace('gnpde::', ''))[:70]
  print('%-70s %8.1f us   gap %6.1f us' % (short, (e - s) / 1e3, (s - prev_end) / 1e3))
  total += e - s
  prev_end = e
print('window: %d kernels, %.1f us of kernel time, %.1f us wall' % (count, total / 1e3, (rows[min(start + count, len(rows)) - 1][1] - rows[start][0]) / 1e3))
