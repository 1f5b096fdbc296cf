#!/bin/bash
# per-kernel time of a short default bench under rocprofv3 --kernel-trace --stats (optionally with VCS_HIP_LIB set)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/stats_quick_$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/run -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/run.log 2>&1
cut -d, -f1-4 $out/run/*/*_kernel_stats.csv | cut -c1-120
