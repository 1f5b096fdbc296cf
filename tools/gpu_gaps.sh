#!/bin/bash
# usage: tools/gpu_gaps.sh <tag> <program args...> -> gpurun_out/gaps_<tag>.txt: idle time between consecutive
# kernels of the steady-state steps (rocprofv3 --kernel-trace timestamps)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/gaps_$tag
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/gaps_$tag -o k -- python3 "$@" > $out/gaps_$tag.log 2>&1
python3 - $out/gaps_$tag <<'PY' | tee $out/gaps_$tag.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
if not f:
    print("no kernel_trace file"); sys.exit(0)
rows = sorted(({'n': r['Kernel_Name'].split('(')[0][:40], 's': int(r['Start_Timestamp']), 'e': int(r['End_Timestamp'])} for r in csv.DictReader(open(f[0]))), key=lambda r: r['s'])
rows = rows[len(rows) // 2:]          # steady state: the second half of the run
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    gap[(a['n'], b['n'])].append(b['s'] - a['e'])
for r in rows: dur[r['n']].append(r['e'] - r['s'])
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("kernel %-42s n %5d avg %8.2f us" % (k, len(v), sum(v) / len(v) / 1e3))
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("gap %-34s -> %-34s n %5d avg %7.2f us  median %7.2f us" % (k[0][:34], k[1][:34], len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3))
PY
grep -E "step|Error|error" $out/gaps_$tag.log | tail -3
