set -u
OUT=gpurun_out/r03b; mkdir -p $OUT
timeout 300 python -m pytest tests/test_tracking_gpu.py -x -q -p no:cacheprovider > $OUT/pytest_tracking.log 2>&1; echo "rc $?" >> $OUT/pytest_tracking.log; tail -15 $OUT/pytest_tracking.log
timeout 200 python tools/tracking_ab.py arxiv 100 > $OUT/tracking_ab_arxiv.log 2>&1; echo "rc $?" >> $OUT/tracking_ab_arxiv.log; tail -8 $OUT/tracking_ab_arxiv.log
timeout 400 python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_tracking_gpu.py > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log; tail -8 $OUT/pytest.log
