o=gpurun_out/t; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$1" > $o/pytest.log 2>&1; rc=$?
tail -12 $o/pytest.log | cut -c1-300
exit $rc
