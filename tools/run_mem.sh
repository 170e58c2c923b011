# memory-hierarchy counters of the band kernels (rocprofv3 --pmc, separate passes): L2 hits / misses / requests / stalls, L1 (TCP / TA) busy and stalls
o=gpurun_out/mem; mkdir -p $o
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
P1="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
P2="TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_LATENCY_FIFO_FULL_sum"
P3="TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE"
P4="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $o/p$i -- python tools/exp_layers.py $o/p$i.json 608 8 autotune=0 > $o/p$i.log 2>&1 || { tail -5 $o/p$i.log; }
  python tools/pmc_summary.py $o/p$i "conv_band_f16s3_kernel<128" > $o/p$i.txt 2>/dev/null
  echo "== pass $i"; head -8 $o/p$i.txt | cut -c1-130
done
find $o -name "*.csv" -size +5M -delete
