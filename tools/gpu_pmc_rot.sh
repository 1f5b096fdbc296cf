#!/bin/bash
# PMC comparison of k_fdct with one resident GOP (warm inputs) and with three in turn
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for sets in 1 3; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    name=$(echo $pass | cut -d' ' -f1)
    out=$root/gpurun_out/pmc_rot/s${sets}_$name
    mkdir -p $out
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out -- python3 $root/bench.py --steps 6 --warmup 3 --no-cpu-baseline --rotate-sets 0 --slot-sets $sets > $out.log 2>&1
  done
done
python3 - $root/gpurun_out/pmc_rot <<'PY'
import csv, glob, sys, collections
base = sys.argv[1]
for sets in (1, 3):
    res = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob(base + '/s%d_*/*/*_counter_collection.csv' % sets):
        for r in csv.DictReader(open(f)):
            if 'k_fdct' not in r['Kernel_Name']: continue
            res[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']].add(r['Dispatch_Id'])
    print('sets', sets, {c: round(v / len(n[c])) for c, v in sorted(res.items())})
PY
