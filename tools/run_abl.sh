# timing-only ablation of the band kernel's main loop (librtod_abl<bits>.so built with -DRTOD_TIMELINE -DRTOD_ABL=<bits>): tools/run_abl.sh "<variants>" "<abl list>"
o=gpurun_out/abl; mkdir -p $o
for v in $1; do
 for abl in $2; do
  RTOD_LIB=$PWD/realtimeobjectdetection_amd/librtod_abl$abl.so timeout -k 10 200 python tools/exp_layers.py $o/v${v}_a$abl.json 608 8 autotune=0 force_f16s3_variant=$v > $o/v${v}_a$abl.log 2>&1
  echo "== variant $v abl $abl"; grep "timeline. band" $o/v${v}_a$abl.log | grep -E "W=38.*res=1|W=76.*res=1" | sort -t'|' -k1,1 -u | cut -c1-250 | head -2
 done
done
