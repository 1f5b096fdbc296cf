#!/bin/bash
# usage: tools/gpu_kstats.sh <tag> <program args...>  -> gpurun_out/kstats_<tag>.txt (per-kernel durations, rocprofv3 --kernel-trace --stats)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/kstats_$tag
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats_$tag -o k -- python3 "$@" > $out/kstats_$tag.log 2>&1
python3 - $out/kstats_$tag <<'PY' | tee $out/kstats_$tag.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
if not f:
    print("no kernel_stats file"); sys.exit(0)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print("%-64s calls %6s avg %9.1f us total %8.2f ms %6s%%" % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
PY
grep -E "step|Error|error" $out/kstats_$tag.log | tail -3
