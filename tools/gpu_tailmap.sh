#!/bin/bash
# A/B of the strip -> workgroup mapping of k_fdct / k_idct (VCS_TAIL_MAP) and of the IEEE
# division sequence against the default exact 3-FMA one (VCS_FAST_DIV=0); bench lines with decode.
export BENCH_EXTRA="--with-decode"
bash tools/gpu_ab.sh "VCS_FAST_DIV=1" "VCS_FAST_DIV=0" "VCS_TAIL_MAP=0" "VCS_TAIL_MAP=1" "VCS_FAST_DIV=1" "VCS_FAST_DIV=0"
