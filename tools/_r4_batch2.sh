python -m pytest tests/test_gpu_linear.py tests/test_gpu_fused_stats.py tests/test_gpu_adain_cached.py -q -x 2>&1 | tail -4
KERNELS=256x256,256x256h python tools/gpu_gemm_stress.py 60 1 2>&1 | grep -v amdgpu.ids | grep "lowpx" | grep -v "0 / 60 runs differ from the first $" | head -20
for shp in "8192 3840 1280" "32768 1920 640" "8192 1920 640" "32768 960 320" "2048 3840 1280"; do for f in 1 0; do SECS=0.5 ROUNDS=3 python tools/_lin_ab_sustained.py $shp 0 $f 256x256 256x256h 2>&1 | grep -v amdgpu.ids; done; done
for shp in "8192 1280 1280" "32768 640 640" "8192 640 640"; do SECS=0.5 ROUNDS=3 python tools/_lin_ab_sustained.py $shp 1 0 256x256 256x256h 128x128 2>&1 | grep -v amdgpu.ids; done
python tools/_find_casts.py 0 2>&1 | grep "adain_affine" | head -1
