#!/bin/bash
# stem2 phase ablation (diagnostic library): time of the fused stem launch with phases switched off (results garbage)
out=$1; mkdir -p $out
for d in ${S2_BITS:-0 1 2 4 8 3 6 7 15}; do
  RTOD_LIB=$PWD/realtimeobjectdetection_amd/librtod_diag.so RTOD_S2_DBG=$d timeout -k 10 120 python tools/exp_layers.py $out/d$d.json 608 8 autotune=0 > /dev/null 2>&1 || exit 1
  python - <<PY
import json
d=json.load(open("$out/d$d.json"))
print("dbg=$d layer1 ms", [r['ms'] for r in d['per_launch'] if r['layer']==1][0])
PY
done
