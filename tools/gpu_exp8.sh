#!/bin/bash
# A/B of experiment builds of the 8x8 search kernel (vcs-h264_amd/lib/exp/*.so, built with -DVCS_EXP_*)
set -o pipefail
mkdir -p gpurun_out
BENCH_EXTRA="--bs 8" bash tools/gpu_ab.sh "VCS_X=0" "VCS_HIP_LIB=vcs-h264_amd/lib/exp/nocore8.so" "VCS_HIP_LIB=vcs-h264_amd/lib/exp/pf2_8.so" \
  "VCS_HIP_LIB=vcs-h264_amd/lib/exp/nobar8.so" "VCS_HIP_LIB=vcs-h264_amd/lib/exp/nobar_nocore8.so" 2>&1 | tee gpurun_out/exp8_ab.log
