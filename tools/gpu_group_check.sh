#!/bin/bash
# parity of the shared-window search kernel (k_me16g) for every group width, then A/B timings
set -o pipefail
mkdir -p gpurun_out
for w in 4 1 2 8; do
  for p in 3 4; do
    VCS_ME16_GROUP=$w VCS_ME16_WPS=$p VCS_ME16_ROLL=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q \
      -k "golden_me_small or golden_me_edges or me_vs_oracle_shapes or static_heavy or 4k_frame or empty_and_tiny or golden_real_small or gop_1080p" \
      -p no:cacheprovider > gpurun_out/group_w${w}_p${p}.log 2>&1 || { echo "PARITY FAIL w=$w p=$p"; tail -30 gpurun_out/group_w${w}_p${p}.log; exit 1; }
    echo "parity ok w=$w wps=$p: $(tail -1 gpurun_out/group_w${w}_p${p}.log)"
  done
done
bash tools/gpu_ab.sh "VCS_ME16_GROUP=0" "VCS_ME16_GROUP=4 VCS_ME16_WPS=2" "VCS_ME16_GROUP=4 VCS_ME16_WPS=3" "VCS_ME16_GROUP=4 VCS_ME16_WPS=4" \
  "VCS_ME16_GROUP=8 VCS_ME16_WPS=3" "VCS_ME16_GROUP=8 VCS_ME16_WPS=4" "VCS_ME16_GROUP=2 VCS_ME16_WPS=3" "VCS_ME16_GROUP=2 VCS_ME16_WPS=4" \
  "VCS_ME16_GROUP=4 VCS_ME16_WPS=4 VCS_ME16G_SEG=34" "VCS_ME16_GROUP=4 VCS_ME16_WPS=4 VCS_ME16G_SEG=23" "VCS_ME16_GROUP=4 VCS_ME16_WPS=4 VCS_ME16G_SEG=12" \
  "VCS_ME16_GROUP=4 VCS_ME16_WPS=3 VCS_ME16G_SEG=34" "VCS_ME16_GROUP=4 VCS_ME16_WPS=3 VCS_ME16G_SEG=12" 2>&1 | tee gpurun_out/group_ab.log
