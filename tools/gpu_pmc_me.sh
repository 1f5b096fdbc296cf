#!/bin/bash
# PMC passes over the default bench for the motion-search kernel: instruction mix, LDS pipe, waits
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-me}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY" \
            "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA" \
            "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/p$i -- python3 $root/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extras $BENCH_EXTRA > $out/p$i.log 2>&1
done
python3 - $out <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(out + '/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        res[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
for k, v in res.items():
    if 'rocclr' in k: continue
    print(k)
    for c, x in sorted(v.items()):
        print("   %-26s %16.0f" % (c, x / len(n[(k, c)])))
PY
