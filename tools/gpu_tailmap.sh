#!/bin/bash
# A/B of the strip -> workgroup mapping of k_fdct / k_idct (VCS_TAIL_MAP) and of the exact
# 3-FMA division (VCS_FAST_DIV); bench lines with the decode leg.
export BENCH_EXTRA="--with-decode"
bash tools/gpu_ab.sh "VCS_TAIL_MAP=0" "VCS_TAIL_MAP=1" "VCS_TAIL_MAP=2" "VCS_FAST_DIV=1" "VCS_TAIL_MAP=0" "VCS_FAST_DIV=1"
