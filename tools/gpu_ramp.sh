#!/bin/bash
# how the measured rate depends on how long the GPU has been busy before / during the timed steps
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 100" "--steps 20 --warmup 500" "--steps 20 --warmup 2000" "--steps 200 --warmup 5" "--steps 2000 --warmup 5" "--steps 20 --warmup 5"; do
  out=$(timeout -k 10 200 python bench.py $cfg --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
  python - "$cfg" "$out" <<'PY'
import sys, json
d = json.loads(sys.argv[2])
print("%-28s value=%9.1f ms/step=%.4f me=%.4f tail=%.4f clock=%s" % (sys.argv[1], d["value"], d["ms_per_step"], d["kernel_ms_per_step"]["me"], d["kernel_ms_per_step"]["enc_tail"], d.get("k_me16g_valu_issue", {}).get("clock_ghz_measured")))
PY
done
