#!/bin/bash
# usage: tools_gpu_ab.sh "ENV=VAL ..." ...   -> one bench line summary per env set
for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $BENCH_EXTRA 2>&1 | tail -1)
  python - "$cfg" "$out" <<'PY'
import sys, json
cfg, out = sys.argv[1], sys.argv[2]
try:
    d = json.loads(out)
    print("%-40s value=%10.1f MB/ms  ms/step=%.4f  kern=%s dec=%s" % (cfg, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d.get("decode")))
except Exception as e:
    print(cfg, "FAILED", out[-300:])
PY
done
