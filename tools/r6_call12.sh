#!/bin/bash
# round 6, GPU call 12: where the host time of the Cora best_params epoch goes (cProfile over bench.py --config cora-epoch)
OUT=gpurun_out/r6c12
mkdir -p $OUT
timeout 600 python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--config', 'cora-epoch', '--steps', '60', '--warmup', '5', '--no-cpu-baseline']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
  runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
  pass
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('cumulative')
ps.print_stats(70)
open('$OUT/cprofile_cumulative.txt', 'w').write(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
open('$OUT/cprofile_tottime.txt', 'w').write(s.getvalue())
" > $OUT/run.log 2>&1
tail -2 $OUT/run.log | cut -c1-400
head -75 $OUT/cprofile_tottime.txt | tail -60 | cut -c1-170
