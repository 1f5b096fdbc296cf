#!/bin/bash
# 8x8 blocks: bit-plane search (VCS_ME8_BITS=2) vs k_me8g / k_me8 (=0) over batch sizes
for n in ${@:-1 2 3 4 8 16 29}; do
  for b in 2 0; do VCS_ME8_BITS=$b python tools/gpu_me_time.py --bs 8 --frames $n --steps 20 --label "1080p 8x8 n=$n BITS=$b"; done
done
