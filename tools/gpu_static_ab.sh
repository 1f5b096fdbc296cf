#!/bin/bash
# content with a static region (the left part of every frame copied from the reference): bit-plane search vs k_me16g
for s in 0.3 0.7 0.95; do
  for b in 1 0; do VCS_ME16_BITS=$b python tools/gpu_me_time.py --static $s --label "static left $s BITS=$b"; done
done
