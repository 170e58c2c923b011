# A/B/n of several libraries on one box, interleaved twice: tools/run_abn.sh "<lib names>" [exp_layers args...]
o=gpurun_out/abn_$(date +%H%M%S); mkdir -p $o; libs=$1; shift
for rep in 1 2; do for L in $libs; do RTOD_LIB=$PWD/realtimeobjectdetection_amd/$L timeout -k 10 200 python tools/exp_layers.py $o/${L%.so}-$rep.json "$@" > /dev/null 2>&1 || echo "$L failed"; done; done
python tools/ab_report.py $o | cut -c1-200
