#!/bin/bash
# time of the 1x1 layer groups under every forced tile variant (autotune off), one box
out=$1; mkdir -p $out
for v in ${SWEEP_V:-auto 0 1 2 3 6 7 9 10 11 70 71 72 73 74 75 76 77}; do
  if [ $v = auto ]; then args=""; else args="autotune=0 force_f16s3_variant=$v"; fi
  timeout -k 10 120 python tools/exp_layers.py $out/v$v.json 608 8 $args > /dev/null 2>&1 || { echo "v$v failed"; continue; }
  python - <<PY
import json, collections
d=json.load(open("$out/v$v.json")); g=collections.defaultdict(float); n=collections.Counter()
for r in d["per_launch"]:
    if r["kind"]==0 and r["k"]==1: g[r["hout"]]+=r["ms"]; n[r["hout"]]+=1
print("v$v", " ".join("%d:%.1fus(x%d)"%(h,1000*g[h]/n[h],n[h]) for h in sorted(g)), "total %.3f"%sum(g.values()), [r["name"] for r in d["per_launch"] if r["layer"]==66][:1])
PY
done
