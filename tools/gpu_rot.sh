#!/bin/bash
# usage: gpu_rot.sh "ENV=VAL ..." ...  -> main and rotating_inputs numbers of the default bench per env set
for cfg in "$@"; do
  env $cfg python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s' % '$cfg'[:28], d['value'], d['ms_per_step'], d['kernel_ms_per_step'], '| rotating', d['rotating_inputs']['value'], d['rotating_inputs']['kernel_ms_per_step'])"
done
