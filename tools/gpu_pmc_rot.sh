#!/bin/bash
# PMC comparison of the bench's kernels with one resident GOP (warm inputs) and with three
# encoded in turn (cold inputs).  The TCC block has FOUR counters per pass: the L2 hit /
# miss / EA request / write-stall counters go in separate passes.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for sets in 1 3; do
  i=0
  for pass in "FETCH_SIZE" "WRITE_SIZE" \
              "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
              "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum" \
              "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum" \
              "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    out=$root/gpurun_out/pmc_rot/s${sets}_p$i
    mkdir -p $out
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out -- python3 $root/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --rotate-sets $sets > $out.log 2>&1 || echo "pass $i (sets $sets) failed: $(tail -2 $out.log)"
  done
done
python3 - $root/gpurun_out/pmc_rot <<'PY' | tee $root/gpurun_out/pmc_rot/summary.txt
import csv, glob, sys, collections
base = sys.argv[1]
for kern in ('k_fdct', 'k_me16g'):
    for sets in (1, 3):
        res = collections.defaultdict(float); n = collections.defaultdict(set)
        for f in glob.glob(base + '/s%d_p*/*/*_counter_collection.csv' % sets):
            for r in csv.DictReader(open(f)):
                if kern not in r['Kernel_Name']: continue
                res[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']].add(r['Dispatch_Id'])
        print(kern, 'resident GOP sets', sets)
        for c, v in sorted(res.items()):
            print('   %-30s %16.0f' % (c, v / len(n[c])))
PY
