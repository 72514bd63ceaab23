"""How evenly does an aggregation launch load the 8 XCDs?  (host only, no GPU)

  python tools/xcd_balance.py [arxiv|rmat|cora ...]            (default: arxiv rmat)

Workgroup b of a launch runs on XCD b % 8 and every XCD walks its own list: every 8th hub chunk (512 entries), then its
share of the rows (csrc/spmm.hip, item_of).  The launch lasts as long as the slowest XCD.  This tool takes the row -> XCD
map from the library itself (gnpde_xcd_row_map: the host + device function the kernels call), for the shipped hashed-block
deal and for contiguous eighths, and prints max / mean of the modelled work per XCD
(entries + 3 per row for rows of <= 512 entries; the hub chunks are dealt round robin in both cases), and which of the two the
graph builder picks for the graph (gnpde_graph_t.xcd_deal: contiguous while that is within 3 % of even).
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnpde_amd as G  # noqa: E402
from gnpde_amd import _lib  # noqa: E402

LONG = 512


def row_map(n, deal):
  L = _lib.lib()
  shift, per = ctypes.c_int32(0), ctypes.c_int32(0)
  _lib.check(L.gnpde_xcd_row_map(0, n, deal, ctypes.byref(shift), ctypes.byref(per), None))
  m = np.full((8, max(per.value, 1)), -1, dtype=np.int32)
  _lib.check(L.gnpde_xcd_row_map(0, n, deal, ctypes.byref(shift), ctypes.byref(per), m.ctypes.data))
  return shift.value, m


def main():
  names = sys.argv[1:] or ['arxiv', 'rmat']
  for name in names:
    ei, n = G.synthetic.make_graph(name, seed=0)
    deg = np.bincount(ei[0].numpy(), minlength=n) + 1          # + the self-loop the functions add
    row_work = np.where(deg <= LONG, deg + 3, 0).astype(np.float64)
    hub_entries = float(deg[deg > LONG].sum())
    out = {'graph': name, 'nodes': int(n), 'entries': int(deg.sum()), 'hub_rows': int((deg > LONG).sum()),
           'hub_entry_fraction': round(hub_entries / deg.sum(), 3)}
    for label, deal in (('hashed_blocks', _lib.XCD_HASHED), ('contiguous_eighths', _lib.XCD_CONTIGUOUS)):
      shift, m = row_map(n, deal)
      rows = np.array([row_work[r[r >= 0]].sum() for r in m])
      total = rows + hub_entries / 8.0
      out[label] = {'block_rows': (1 << shift) if shift >= 0 else None,
                    'row_entries_per_xcd': [int(deg[r[r >= 0]][deg[r[r >= 0]] <= LONG].sum()) for r in m],
                    'rows_max_over_mean': round(float(rows.max() / rows.mean()), 4),
                    'launch_max_over_mean': round(float(total.max() / total.mean()), 4)}
    g = G.CSRGraph(G.add_remaining_self_loops(ei, None, 1.0, n)[0], n, device='cpu')
    out['graph_builder_picks'] = 'hashed_blocks' if g.struct.xcd_deal == _lib.XCD_HASHED else 'contiguous_eighths'
    out['graph_builder_measured_contiguous_imbalance'] = round(g.xcd_imbalance_contiguous, 4)
    print(json.dumps(out))


if __name__ == '__main__':
  main()
