"""Worker for tests/test_distributed_cpu.py::test_transport_ladder_falls_down_every_rung (torch.distributed.run, gloo, CPU).

The negotiation bench.py --gpus N uses to pick its halo transport (distributed.negotiate_transport: p2p -> rccl -> torch, a rung is
taken only if EVERY rank gets through both of its checks) with scripted per-rank failures in place of the device work: whichever
rank fails, at whichever phase, all ranks must drop that rung together, release what it set up, and agree on the next one."""
import json
import sys

import torch.distributed as dist

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gnpde_amd import distributed as D  # noqa: E402


def main():
  out_path = sys.argv[1]
  dist.init_process_group('gloo')
  rank, world = dist.get_rank(), dist.get_world_size()

  def agree(ok):
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    return all(flags)

  # scenario -> {(candidate, phase, rank): failure}; 'raise' = the attempt throws, 'bad' = it reports a wrong result
  scenarios = {
    'all_fine': {},
    'p2p_fails_on_rank1_phase1': {('p2p', 0, 1): 'raise'},
    'p2p_fails_on_rank0_phase2': {('p2p', 1, 0): 'bad'},
    'p2p_and_rccl_fail': {('p2p', 0, 0): 'raise', ('rccl', 1, world - 1): 'raise'},
    'everything_fails': {('p2p', 0, 1): 'bad', ('rccl', 0, 0): 'bad', ('torch', 1, 1): 'raise'},
    'only_torch_offered': {},
  }
  expected = {'all_fine': 'p2p', 'p2p_fails_on_rank1_phase1': 'rccl', 'p2p_fails_on_rank0_phase2': 'rccl', 'p2p_and_rccl_fail': 'torch',
              'everything_fails': None, 'only_torch_offered': 'torch'}
  res = {}
  for name, fails in scenarios.items():
    ladder = ['torch'] if name == 'only_torch_offered' else ['p2p', 'rccl', 'torch']
    calls, rejected, notes = [], [], {}

    def make_phase(k):
      def phase(cand):
        calls.append((cand, k))
        mode = fails.get((cand, k, rank))
        if mode == 'raise':
          raise RuntimeError('scripted failure of %s in phase %d on rank %d' % (cand, k + 1, rank))
        if mode == 'bad':
          return False, 'scripted wrong result of %s in phase %d on rank %d' % (cand, k + 1, rank)
        return True, None
      return phase
    chosen = D.negotiate_transport(ladder, [make_phase(0), make_phase(1)], agree, notes, on_reject=rejected.append)
    everyone = [None] * world
    dist.all_gather_object(everyone, (chosen, rejected))
    res[name] = dict(chosen=chosen, expected=expected[name], same_on_all_ranks=all(e == everyone[0] for e in everyone), rejected=rejected,
                     notes=notes, calls=calls)
    dist.barrier()
  allres = [None] * world
  dist.all_gather_object(allres, res)
  if rank == 0:
    json.dump(allres, open(out_path, 'w'))
  dist.destroy_process_group()
  print('LADDER_OK')


if __name__ == '__main__':
  main()
