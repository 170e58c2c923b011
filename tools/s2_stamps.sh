#!/bin/bash
# stem2 per-phase s_memtime attribution (diagnostic library)
RTOD_LIB=$PWD/realtimeobjectdetection_amd/librtod_diag.so RTOD_S2_STAMPS=1 RTOD_S2_DBG=64 timeout -k 10 120 python tools/exp_layers.py $1/stamps.json 608 8 autotune=0 2>&1 | grep -E "s2 stamps|sum"
