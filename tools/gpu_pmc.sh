#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> "<counters...>" <program args...> -> gpurun_out/pmc_<tag>.txt (per-kernel counter sums per dispatch)
tag=$1; ctrs=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/pmc_$tag
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc_$tag -o p -- python3 "$@" > $out/pmc_$tag.log 2>&1
python3 - $out/pmc_$tag <<'PY' | tee $out/pmc_$tag.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0][-40:]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
for k, v in acc.items():
    n = len(disp[k])
    print(k, "dispatches", n)
    for c, x in sorted(v.items()):
        print("   %-28s %16.0f per dispatch" % (c, x / n))
PY
