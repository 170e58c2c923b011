set -e
o=gpurun_out/r3n; mkdir -p $o
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $o/trace -- python tools/exp_gaps.py run 608 8 10 > $o/run.log 2>&1
python tools/exp_gaps.py report $o/trace | tee $o/gaps.txt
find $o -name "*.csv" -size +5M -delete
