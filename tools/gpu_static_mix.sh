#!/bin/bash
for kind in "region 0.7" "random 0.7" "region 0.3" "random 0.9"; do
  for cfg in "VCS_ME16G_DYN=0" "VCS_ME16G_DYN=1" "VCS_ME16G_DYN=1 VCS_ME16G_SEG=9" "VCS_ME16G_DYN=0 VCS_ME16G_SEG=9"; do
    env $cfg timeout -k 10 100 python tools/gpu_static_mix.py $kind || exit 1
  done
done
