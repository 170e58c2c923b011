# A/B/n of several libraries on one box, interleaved twice: tools/run_abn.sh "<lib names>" [exp_layers args...]
# (RTOD_TILES=profiles/r03_tiles.json in the environment: every library runs the same tile table instead of autotuning)
o=gpurun_out/abn_$(date +%H%M%S); mkdir -p $o; libs=$1; shift
for rep in $(seq 1 ${REPS:-2}); do for L in $libs; do RTOD_LIB=$PWD/realtimeobjectdetection_amd/$L timeout -k 10 200 python tools/exp_layers.py $o/${L%.so}-$rep.json "$@" > $o/${L%.so}-$rep.log 2>&1 || { echo "$L failed"; tail -3 $o/${L%.so}-$rep.log; }; done; done
python tools/ab_report.py $o | cut -c1-200
