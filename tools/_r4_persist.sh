#!/usr/bin/env bash
# persistent (one workgroup per CU) vs one workgroup per tile, on the step's tiled GEMM shapes (round 4 re-check incl. the short-K shapes)
for shp in "32768 1920 640 0 1" "8192 3840 1280 0 1" "32768 960 320 0 1" "8192 1920 640 0 1" "32768 640 640 1 0" "8192 1280 1280 1 0" "32768 320 320 1 0"; do
  for pm in 0 1; do
    export IR_LIN_PERSISTENT=$pm   # 1 (the default since round 4) / 0
    echo -n "persistent=$pm  "; SECS=0.5 ROUNDS=3 python tools/_lin_ab_sustained.py $shp 256x256 2>&1 | grep -v amdgpu.ids
  done
done
