"""Wave-slot model of the second row-attention launch (host only): rows of 17..512 entries, one wavefront each, the hub-chunk
blocks of phase 1 riding first.  A row costs a fixed part plus two dependent round trips per 64 entries; items are handed to
the earliest free of 8 192 wave slots in list order.  Compares the row order of the record list (until round-2 v3) with the
longest-first order the graph builders now produce, with the chunk-partial fold read from memory or staged through LDS.
One free parameter, the round-trip time; 0.85 us reproduces the measured 30 us of the ogbn-arxiv shape.

  python tools/attention_slot_model.py [arxiv|rmat|...] [round_trip_us]
"""
import heapq
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnpde_amd as G  # noqa: E402

LONG, SLOTS = 512, 8192


def makespan(items, slots=SLOTS):
  free = [0.0] * slots
  heapq.heapify(free)
  end = 0.0
  for d in items:
    t = heapq.heappop(free) + d
    heapq.heappush(free, t)
    end = max(end, t)
  return end


def main():
  name = sys.argv[1] if len(sys.argv) > 1 else 'arxiv'
  rt = float(sys.argv[2]) if len(sys.argv) > 2 else 0.85
  ei, n = G.synthetic.make_graph(name, seed=0)
  deg = np.bincount(ei[0].numpy(), minlength=n) + 1
  rows = deg[(deg > 16) & (deg <= LONG)]
  hubs = deg[deg > LONG]
  chunks = np.ceil(hubs / float(LONG)).astype(np.int64)
  row_cost = 1.5 + rt + 2.0 * rt * np.ceil(rows / 64.0)
  out = {'graph': name, 'round_trip_us': rt, 'rows_17_to_512': int(rows.size), 'rows_over_64': int((rows > 64).sum()),
         'hub_chunks': int(chunks.sum()), 'largest_hub_chunks': int(chunks.max()) if chunks.size else 0}
  for fold_name, per_chunk in (('fold_from_memory', rt), ('fold_staged_in_lds', 0.0)):
    # every chunk block folds its row's chunk partials: ~2 dependent loads per chunk of the row from memory, one round trip staged
    fold = [1.5 + (2 * c * per_chunk / 2.0 if per_chunk else rt) + 2 * rt for c in np.repeat(chunks, chunks)]
    hub_items = [f for f in fold for _ in range(4)]          # a block = 4 wavefront slots
    out[fold_name] = {'row_order_us': round(makespan(hub_items + list(row_cost)), 1),
                      'longest_first_us': round(makespan(hub_items + sorted(row_cost, reverse=True)), 1)}
  print(json.dumps(out))


if __name__ == '__main__':
  main()
