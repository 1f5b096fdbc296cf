#!/bin/bash
# one short PMC pass over the default bench: VALU / LDS instruction counts and bank conflicts per dispatch
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_quick
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $out/run -- python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/run.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(out + '/run/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        res[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
for k, v in res.items():
    if 'rocclr' in k: continue
    print(k, {c: round(x / len(n[(k, c)])) for c, x in v.items()})
PY
