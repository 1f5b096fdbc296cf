#!/bin/bash
# bit-plane search (BITS=2: forced) vs k_me16g (BITS=0) at 4K over batch sizes
for n in 4 6 8 12 15 29; do
  for b in 2 0; do VCS_ME16_BITS=$b python tools/gpu_me_time.py --H 2160 --W 3840 --frames $n --sets 1 --steps 8 --label "4K n=$n BITS=$b"; done
done
