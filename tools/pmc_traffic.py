"""Fold rocprofv3 --pmc passes into the per-kernel traffic record bench.py reads (profiles/hbm_traffic.json).

  python tools/pmc_traffic.py OUT.json KEY COMMIT CMD  DIR [DIR ...]

Each DIR is the output directory of ONE rocprofv3 --pmc pass (separate passes: FETCH_SIZE | WRITE_SIZE TCC_HIT_sum
TCC_MISS_sum | TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum ..., MI355X_MICROARCH.md "rocprofv3 PMC slots") of the same
command.  Per kernel: mean counter values per launch, and
    bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE is reported in KiB and, on gfx950, counts 64 B per 128-B request of a wide coalesced read, so it is doubled as
MI355X_MICROARCH.md section HBM prescribes; WRITE_SIZE is taken as reported).  The record of the dominant aggregation kernel
is stored under KEY (e.g. arxiv_d128_spmm) together with its provenance (kernel name, commit, command).
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
  name = re.sub(r'\(anonymous namespace\)::', '', name)
  name = re.sub(r'^void ', '', name)
  return name.split('(')[0]


def main():
  out_path, key, commit, cmd = sys.argv[1:5]
  dirs = sys.argv[5:]
  acc = {}
  for d in dirs:
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
      for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        if not k.startswith('gnpde::'):
          continue
        ent = acc.setdefault(k, {}).setdefault(r['Counter_Name'], [0, 0.0])
        ent[0] += 1
        ent[1] += float(r['Counter_Value'])
  detail = {}
  for k, cs in sorted(acc.items()):
    m = {c: v[1] / v[0] for c, v in cs.items()}
    rec = {'launches': max(v[0] for v in cs.values()), 'counters_mean_per_launch': {c: round(x, 3) for c, x in m.items()}}
    if 'FETCH_SIZE' in m:
      rec['fetch_bytes'] = 2.0 * m['FETCH_SIZE'] * 1024
    if 'WRITE_SIZE' in m:
      rec['write_bytes'] = m['WRITE_SIZE'] * 1024
    if 'fetch_bytes' in rec and 'write_bytes' in rec:
      rec['bytes_per_launch'] = rec['fetch_bytes'] + rec['write_bytes']
    if 'TCC_HIT_sum' in m and 'TCC_MISS_sum' in m and m['TCC_HIT_sum'] + m['TCC_MISS_sum'] > 0:
      rec['l2_hit_rate'] = round(m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']), 4)
    # (TCC_EA0_RDREQ_DRAM_sum equals TCC_EA0_RDREQ_sum on gfx950 whether or not the Infinity Cache serves the request, so it
    #  does not separate MALL hits from DRAM reads; the raw values stay in counters_mean_per_launch)
    if 'TCP_UTCL1_TRANSLATION_MISS_sum' in m and m.get('TCP_UTCL1_REQUEST_sum', 0) > 0:
      rec['utcl1_miss_rate'] = round(m['TCP_UTCL1_TRANSLATION_MISS_sum'] / m['TCP_UTCL1_REQUEST_sum'], 5)
    detail[k] = rec
  # dominant aggregation kernel = the spmm kernel with the most fetched bytes
  agg = [k for k in detail if re.search(r'spmm_(rows|wide|pair)_kernel', k) and 'bytes_per_launch' in detail[k]]
  data = {}
  if os.path.exists(out_path):
    try:
      data = json.load(open(out_path))
    except Exception:
      data = {}
  data['_note'] = ('HBM/fabric bytes per launch from rocprofv3 --pmc, separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 '
                   '(FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B; WRITE_SIZE as reported). '
                   'Written by tools/pmc_traffic.py; every record carries the commit and command it was measured with.')
  if agg:
    k = max(agg, key=lambda n: detail[n]['bytes_per_launch'] * detail[n]['launches'])
    rec = dict(detail[k])
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for rel in ('graph-neural-pde_amd/csrc/spmm.hip', 'graph-neural-pde_amd/csrc/attention.hip', 'graph-neural-pde_amd/csrc/linear.hip',
                'graph-neural-pde_amd/csrc/epilogue.h'):     # = bench.KERNEL_SOURCES: the record is stale when ANY of them changes
      h.update(open(os.path.join(root, rel), 'rb').read())
    rec.update(kernel=k, commit=commit, command=cmd, method='rocprofv3 --pmc, mean over the launches of the command',
               kernel_sources_sha16=h.hexdigest()[:16], node_order=os.environ.get('GNPDE_REORDER', 'auto'))
    data[key] = rec
  data.setdefault('detail', {})[key] = {'commit': commit, 'command': cmd, 'kernels': detail}
  json.dump(data, open(out_path, 'w'), indent=1)
  print(json.dumps({key: data.get(key)}, indent=1))


if __name__ == '__main__':
  main()
