#!/bin/bash
# Run ON THE GPU BOX (through gpurun): bench line + per-launch table, rocprofv3 kernel stats, HBM traffic counters.
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{bench.log,layers.json,stats/,fetch/,write/,sq/}
# Counters are collected in their own runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains beside --pmc).
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
set -e
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --layers-out $out/layers.json > $out/bench.log 2> $out/bench.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --res 416 --no-cpu-baseline > $out/bench_416.log 2>> $out/bench.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --precision fp32 --no-cpu-baseline > $out/bench_fp32.log 2>> $out/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $out/stats.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-roofline > $out/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-roofline > $out/write.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/sq -- python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-roofline > $out/sq.log 2>&1
# keep the merge small: the per-dispatch CSVs are large, the summaries are what gets committed
find $out -name "*kernel_trace.csv" -size +20M -delete
echo profile_round done
