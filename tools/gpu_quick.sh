#!/bin/bash
# parity (search kernels, default + forced-group variants) then the two headline timings
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -p no:cacheprovider > gpurun_out/quick_parity.log 2>&1 || { echo "PARITY FAIL"; tail -30 gpurun_out/quick_parity.log; exit 1; }
echo "parity: $(tail -1 gpurun_out/quick_parity.log)"
bash tools/gpu_ab.sh "VCS_X=0" "VCS_X=1" 2>&1 | tee gpurun_out/quick_ab.log
BENCH_EXTRA="--bs 8" bash tools/gpu_ab.sh "VCS_X=0" "VCS_X=1" 2>&1 | tee -a gpurun_out/quick_ab.log
