#!/bin/bash
# Infinity Cache prefetch variants: parity, then 0 = off, 1 = inside the search kernel, 2 = separate pass
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_multi.py -m gpu -x -q -p no:cacheprovider > gpurun_out/pf_parity.log 2>&1 || { echo "PARITY FAIL"; tail -30 gpurun_out/pf_parity.log; exit 1; }
echo "parity: $(tail -1 gpurun_out/pf_parity.log)"
bash tools/gpu_ab.sh "VCS_L3_PREFETCH=0" "VCS_L3_PREFETCH=1" "VCS_L3_PREFETCH=2" "VCS_L3_PREFETCH=0" "VCS_L3_PREFETCH=1" "VCS_L3_PREFETCH=2" 2>&1 | tee gpurun_out/pf_ab.log
BENCH_EXTRA="--strong" bash tools/gpu_ab.sh "VCS_L3_PREFETCH=1" "VCS_L3_PREFETCH=2" 2>&1 | tee -a gpurun_out/pf_ab.log
BENCH_EXTRA="--group" bash tools/gpu_ab.sh "VCS_L3_PREFETCH=1" "VCS_L3_PREFETCH=2" 2>&1 | tee -a gpurun_out/pf_ab.log
