#!/bin/bash
# two bench ranks on one GPU (RCCL refuses duplicate devices): the no-collective fall-back
mkdir -p gpurun_out
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29671 VCS_BENCH_DEVICE=0 VCS_RENDEZVOUS_TAG=try$$ \
    timeout -k 10 120 python bench.py --gpus 2 --steps 2 --warmup 1 --height 256 --width 320 --frames 3 --no-extras --no-cpu-baseline \
    > gpurun_out/fb_$r.out 2> gpurun_out/fb_$r.err &
done
wait
for r in 0 1; do echo "== rank $r"; cat gpurun_out/fb_$r.out; grep -v "^RCCL\|^$" gpurun_out/fb_$r.err | tail -15; done
