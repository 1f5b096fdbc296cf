#!/bin/bash
# k_me8g: LDS row pitch (16-byte slots per window row) vs occupancy and bank conflicts; parity first
set -o pipefail
mkdir -p gpurun_out
for n in 13 15 17 19 20; do
  VCS_HIP_LIB=vcs-h264_amd/lib/exp/slots$n.so VCS_ME16_ROLL=2 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q \
      -k "golden_me_small or golden_me_edges or me_vs_oracle_shapes or golden_real_small or golden_me_1080p" -p no:cacheprovider > gpurun_out/slots_parity.log 2>&1 || { echo "PARITY FAIL slots=$n"; tail -20 gpurun_out/slots_parity.log; exit 1; }
  echo "parity ok slots=$n: $(tail -1 gpurun_out/slots_parity.log)"
done
BENCH_EXTRA="--bs 8" bash tools/gpu_ab.sh "VCS_X=24" "VCS_HIP_LIB=vcs-h264_amd/lib/exp/slots13.so" "VCS_HIP_LIB=vcs-h264_amd/lib/exp/slots15.so" "VCS_HIP_LIB=vcs-h264_amd/lib/exp/slots17.so" "VCS_HIP_LIB=vcs-h264_amd/lib/exp/slots19.so" "VCS_HIP_LIB=vcs-h264_amd/lib/exp/slots20.so" "VCS_X=24" "VCS_L3_PREFETCH=2" "VCS_L3_PREFETCH=0" 2>&1 | tee gpurun_out/slots_ab.log
