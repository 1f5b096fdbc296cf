#!/bin/bash
# kernel times of the product build and of the experiment builds under tools/exp/ (gpu_me_time.py)
set -e
python tools/gpu_me_time.py --label product
VCS_ME16_BITS=0 python tools/gpu_me_time.py --label "product BITS=0 (k_me16g)"
for l in tools/exp/*.so; do VCS_HIP_LIB=$PWD/$l python tools/gpu_me_time.py; done
