#!/bin/bash
for r in 1 2 3 5 8; do for w in 256 512 1024; do VCS_BITS_PF_RATIO=$r VCS_BITS_PF_WGS=$w python tools/gpu_me_time.py --label "ratio=$r wgs=$w"; done; done
