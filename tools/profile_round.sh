#!/bin/bash
# Run ON THE GPU BOX (through gpurun): bench line + per-launch table, rocprofv3 kernel stats, HBM traffic counters.
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{bench.log,layers.json,tiles.json,stats/,fetch/,write/,sq/}
# The first bench run autotunes and writes its tile table (tiles.json); every profiled pass installs that table
# (bench.py --tiles: rtod_plan_set_tiles), so all passes launch exactly the kernels of the timing run and no autotune launch is
# counted.  Counters are collected in their own runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains beside --pmc).
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
T=$out/tiles.json
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
set -e
rm -f $T   # never replay a tile table measured on an older build
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --tiles $T --layers-out $out/layers.json > $out/bench.log 2> $out/bench.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --res 416 --no-cpu-baseline --no-extras > $out/bench_416.log 2>> $out/bench.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --precision fp32 --no-cpu-baseline --no-extras > $out/bench_fp32.log 2>> $out/bench.err
P="--no-cpu-baseline --no-roofline --no-extras --tiles $T"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 20 --warmup 5 $P > $out/stats.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python bench.py --steps 3 --warmup 1 --inflight 1 $P > $out/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python bench.py --steps 3 --warmup 1 --inflight 1 $P > $out/write.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/sq -- python bench.py --steps 3 --warmup 1 --inflight 1 $P > $out/sq.log 2>&1
# keep the merge small: the per-dispatch CSVs are large, the summaries are what gets committed
find $out -name "*kernel_trace.csv" -size +20M -delete
echo profile_round done
