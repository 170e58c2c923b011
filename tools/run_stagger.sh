# start-stagger sweep of the conv_bandd tiles (RTOD_BD_STAGGER: units of 1024 cycles per workgroup slot): tools/run_stagger.sh "<values>" <variants...>
o=gpurun_out/stagger; mkdir -p $o; vals=$1; shift
for rep in 1 2; do for v in $vals; do
  RTOD_BD_STAGGER=$v timeout -k 10 200 python tools/exp_band_modes.py $o/s$v-$rep.json 608 8 "$@" > $o/s$v-$rep.log 2>&1 || { echo "stagger $v failed"; tail -3 $o/s$v-$rep.log; }
  echo "== stagger $v rep $rep"; grep variant $o/s$v-$rep.log | cut -c1-230
done; done
