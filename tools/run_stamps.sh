# In-kernel phase stamps of the band kernel (make -C realtimeobjectdetection_amd/csrc stamps first):  tools/run_stamps.sh [outdir]
o=${1:-gpurun_out/stamps}; mkdir -p $o
RTOD_LIB=$PWD/realtimeobjectdetection_amd/librtod_stamps.so timeout -k 10 300 python tools/exp_layers.py $o/layers.json 608 8 autotune=0 > $o/run.log 2>&1; rc=$?
grep "\[stamps\]" $o/run.log | sort | uniq -c | sort -rn | head -5 > /dev/null
grep "\[stamps\]" $o/run.log | tail -60 > $o/stamps_tail.log
tail -3 $o/run.log
exit $rc
