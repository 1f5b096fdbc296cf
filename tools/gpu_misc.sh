#!/bin/bash
# one-line summaries of non-default bench shapes
for args in "--bs 8" "--bs 8 --frames 1" "--height 2160 --width 3840 --frames 8" "--frames 1"; do
  out=$(timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $args 2>&1 | tail -1)
  python - "$args" "$out" <<'PY'
import sys, json
a, out = sys.argv[1], sys.argv[2]
try:
    d = json.loads(out); print("%-45s value=%10.1f MB/ms ms/step=%.4f kern=%s searched=%s" % (a, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["config"]["searched_block_fraction"]))
except Exception as e:
    print(a, "FAILED", out[-300:])
PY
done
