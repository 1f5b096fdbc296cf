#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers.  Run ON THE GPU BOX
# from the repo root (gpurun -- 'bash profiles/collect.sh r01'); copies of the
# summaries are then committed under profiles/.
#   1. --kernel-trace --stats of the default bench command (per-kernel time)
#   2. PMC passes, each in its own run (FETCH_SIZE and WRITE_SIZE cannot share a
#      pass, the TCC block holds four counters; never combined with sys/hip/hsa tracing)
# Afterwards, here: cp gpurun_out/profiles_<tag>/{pmc_per_dispatch.json,kernel_stats.csv} profiles/<tag>/
# and python profiles/make_traffic.py profiles/<tag>/pmc_per_dispatch.json -> profiles/hbm_traffic.json
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# --with-decode: k_idct over the same GOP sets behind the timed loop (its kernel-trace average and its
# FETCH_SIZE / WRITE_SIZE land in the same files)
B="$root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --with-decode"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $B > $out/stats.log 2>&1
cp $out/stats/*/*_kernel_stats.csv $out/kernel_stats.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
            "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc_$name -- python3 $B > $out/pmc_$name.log 2>&1
done
python3 - $out <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(out + '/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        res[k][r['Counter_Name']] += float(r['Counter_Value']); disp[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
summary = {}
for k, v in res.items():
    if 'rocclr' in k: continue
    summary[k] = {c: x / len(disp[(k, c)]) for c, x in v.items()}
    summary[k]['dispatches_sampled'] = max(len(disp[(k, c)]) for c in v)
# the library these counters belong to (bench.py reports roofline.traffic only for this build)
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'vcs-h264_amd'))
import vcs_hip
summary['build_id'] = vcs_hip.build_id()
json.dump(summary, open(out + '/pmc_per_dispatch.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
PY
cat $out/kernel_stats.csv
grep "^{\"metric\"" $out/stats.log | tail -1 > $out/bench_line_under_rocprof.json; cat $out/bench_line_under_rocprof.json
