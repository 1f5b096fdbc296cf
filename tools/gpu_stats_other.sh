#!/bin/bash
# rocprofv3 --kernel-trace --stats of the other bench shapes (8x8 blocks, config 4 on one GPU, the RCCL path with one GPU)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/stats_other
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in "bs8:--bs 8" "strong4k:--strong" "group:--group"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $args > $out/$tag.log 2>&1
  cp $out/$tag/*/*_kernel_stats.csv $out/kernel_stats_$tag.csv
  echo "== $tag"; cut -d, -f1-4 $out/kernel_stats_$tag.csv | cut -c1-150 | head -8
done
