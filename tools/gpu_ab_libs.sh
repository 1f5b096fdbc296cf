#!/bin/bash
# A/B of library builds on ONE box: tools/gpu_ab_libs.sh [reps] -- every tools/exp/*.so, interleaved
reps=${1:-2}; shift
for r in $(seq $reps); do for l in tools/exp/*.so; do VCS_HIP_LIB=$PWD/$l python tools/gpu_me_time.py --label "$(basename $l) #$r" "$@"; done; done
