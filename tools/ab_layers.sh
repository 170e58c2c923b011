#!/bin/bash
# A/B of two builds of librtod on one box: per-launch tables, alternating A B A B (tools/exp_layers.py); usage: ab_layers.sh outdir libA libB [exp_layers args]
out=$1; A=$2; B=$3; shift 3
mkdir -p $out
for rep in 1 2; do
  RTOD_LIB=$A timeout -k 10 120 python tools/exp_layers.py $out/A$rep.json "$@" || exit 1
  RTOD_LIB=$B timeout -k 10 120 python tools/exp_layers.py $out/B$rep.json "$@" || exit 1
done
