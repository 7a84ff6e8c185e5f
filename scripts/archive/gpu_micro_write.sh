#!/bin/bash
# file-write strategies on the GPU box's /tmp (scripts/micro/write_rate.c)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/micro
{
  df -h /tmp /dev/shm; nproc; free -g; cat /sys/fs/cgroup/cpu.max 2>/dev/null; uname -r
  gcc -O2 -o /tmp/write_rate scripts/micro/write_rate.c -lpthread
  for T in 4 8 16; do /tmp/write_rate /tmp 4 $T; done
  /tmp/write_rate /dev/shm 4 16
} > gpurun_out/micro/write_rate.log 2>&1
cat gpurun_out/micro/write_rate.log
