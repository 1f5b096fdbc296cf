#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for w in 4 1 2 8; do
VCS_ME8_GROUP=$w VCS_ME16_ROLL=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q \
  -k "golden_me_small or golden_me_edges or me_vs_oracle_shapes or static_heavy or empty_and_tiny or golden_real_small or golden_me_1080p or gop_matches_single or config1 or codec_dropin or encode_decode_vs_oracle" \
  -p no:cacheprovider > gpurun_out/q8_parity.log 2>&1 || { echo "PARITY FAIL w=$w"; tail -30 gpurun_out/q8_parity.log; exit 1; }
echo "parity ok w=$w: $(tail -1 gpurun_out/q8_parity.log)"
done
BENCH_EXTRA="--bs 8" bash tools/gpu_ab.sh "VCS_X=1" "VCS_X=2" "VCS_ME8_WPS=4" 2>&1 | tee gpurun_out/q8_ab.log
