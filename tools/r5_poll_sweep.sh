#!/bin/bash
# Development: the single-XCD persistent recurrences under poll knobs (dev builds: tools/lstm_dev_build.sh <tag> -DLC_P_FIRSTLOOK=n / -DLC_P_BACKOFF=n)
out=gpurun_out/${1:-r5g}; mkdir -p $out
for tag in base fl8 fl16 fl24 fl32 bo4 bo8 fl16bo4; do
  [ $tag = base ] && unset LC_DEV_LIB || export LC_DEV_LIB=$tag
  echo "== $tag bf16"; BF16=1 timeout 300 python tools/persist_probe.py 2>&1 | grep -v amdgpu
done > $out/poll_sweep_bf16.txt 2>&1
for tag in base fl8 fl16 fl24; do
  [ $tag = base ] && unset LC_DEV_LIB || export LC_DEV_LIB=$tag
  echo "== $tag f32"; timeout 300 python tools/persist_probe.py 2>&1 | grep -v amdgpu
  echo "== $tag x3"; X3=1 timeout 300 python tools/persist_probe.py 2>&1 | grep -v amdgpu
done > $out/poll_sweep_f32_x3.txt 2>&1
tail -50 $out/poll_sweep_bf16.txt
