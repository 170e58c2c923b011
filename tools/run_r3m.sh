o=gpurun_out/r3m; mkdir -p $o
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; rc=$?
tail -8 $o/pytest.log
exit $rc
