#!/bin/bash
# 1x1 layer groups with operand loads / epilogue switched off (diagnostic library; results garbage): where do their ~20 us go?
out=$1; mkdir -p $out
for z in 0 1 2 3 4 7; do
  RTOD_LIB=$PWD/realtimeobjectdetection_amd/librtod_diag.so RTOD_DBG_ZERO=$z timeout -k 10 120 python tools/exp_layers.py $out/z$z.json 608 8 autotune=0 > /dev/null 2>&1 || { echo "z$z failed"; continue; }
  python - <<PY
import json, collections
d=json.load(open("$out/z$z.json")); g=collections.defaultdict(float); n=collections.Counter()
for r in d["per_launch"]:
    if r["kind"]==0 and r["k"]==1 and r["cout"]%8==0: g[r["hout"]]+=r["ms"]; n[r["hout"]]+=1
print("zero=$z (1: no A loads, 2: no B loads, 4: no epilogue)", " ".join("%d:%.1fus(x%d)"%(h,1000*g[h]/n[h],n[h]) for h in sorted(g)))
PY
done
