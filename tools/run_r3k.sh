set -e
o=gpurun_out/r3k; mkdir -p $o
timeout -k 10 600 python tools/dbg_trainbn.py $o/trainbn_layers.json > $o/trainbn.log 2>&1 || (tail -20 $o/trainbn.log; exit 1)
cat $o/trainbn.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "training_mode or every_tile_variant or launch_kernel" > $o/pytest_sel.log 2>&1 || (tail -30 $o/pytest_sel.log; exit 1)
tail -5 $o/pytest_sel.log
