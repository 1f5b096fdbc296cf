#!/bin/bash
# PMC comparison of the search kernel between library builds: tools/gpu_pmc_ab.sh lib1.so lib2.so ...
# (two counter passes per build over tools/gpu_me_time.py; per-dispatch averages of the kernels matching $KPAT)
root=${GRAFT_REPO_ROOT:-$(pwd)}
KPAT=${KPAT:-k_me16b}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  out=$root/gpurun_out/pmcab_$tag
  rm -rf $out; mkdir -p $out
  i=0
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
              "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    VCS_HIP_LIB=$root/$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/p$i -- python3 $root/tools/gpu_me_time.py --steps 8 > $out/p$i.log 2>&1
  done
  python3 - $out $tag $KPAT <<'PY'
import csv, glob, sys, collections
out, tag, pat = sys.argv[1:4]
res = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(out + '/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if pat not in r['Kernel_Name']: continue
        res[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']].add(r['Dispatch_Id'])
print(tag, pat)
for c, x in sorted(res.items()):
    print("   %-26s %16.0f" % (c, x / len(n[c])))
PY
done
