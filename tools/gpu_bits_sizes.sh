#!/bin/bash
# bit-plane search vs the shared-window kernel over batch sizes and frame sizes (tools/gpu_me_time.py)
for n in 1 2 4 8 16 29; do
  for b in 2 0; do VCS_ME16_BITS=$b python tools/gpu_me_time.py --frames $n --steps 20 --label "1080p n=$n BITS=$b"; done
done
for n in 4 29; do
  for b in 2 0; do VCS_ME16_BITS=$b python tools/gpu_me_time.py --H 2160 --W 3840 --frames $n --sets 1 --steps 6 --label "4K n=$n BITS=$b"; done
done
