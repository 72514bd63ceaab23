"""Print / save a trimmed rocprofv3 kernel_stats CSV (kernel names shortened)."""
import csv, re, sys
src, dst, note = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None), (sys.argv[3] if len(sys.argv) > 3 else '')
rows = list(csv.DictReader(open(src)))
lines = ['# ' + note, 'name,calls,total_ns,avg_ns,pct,min_ns,max_ns']
for r in rows:
  name = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
  name = re.sub(r'^void ', '', name)
  if len(name) > 90:
    name = name.split('(')[0]
  name = name[:100].replace(',', ';')
  lines.append('%s,%s,%s,%.0f,%s,%s,%s' % (name, r['Calls'], r['TotalDurationNs'], float(r['AverageNs']), r['Percentage'], r['MinNs'], r['MaxNs']))
if dst:
  open(dst, 'w').write('\n'.join(lines) + '\n')
for l in lines[:14]:
  print(l)
