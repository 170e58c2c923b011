# A/B of two libraries on one box, interleaved: tools/run_ab.sh <libA> <libB> [exp_layers args...]
o=gpurun_out/ab_$(date +%H%M%S); A=$1; B=$2; shift 2
bash tools/ab_layers.sh $o $PWD/realtimeobjectdetection_amd/$A $PWD/realtimeobjectdetection_amd/$B "$@" > /dev/null 2>&1 || { echo "ab_layers failed"; tail -5 $o/*.log 2>/dev/null; exit 1; }
python tools/ab_report.py $o
