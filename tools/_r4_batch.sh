set -x
python tools/gpu_time.py 0,11,12,13,19 8 4 512 presc 2>&1 | grep -v amdgpu.ids | tee gpurun_out/layer_classes_r4a.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "w64x4 or w64x8 or default" 2>&1 | tail -4
for sh in top capture l1024 l1024cap; do python tools/gpu_energy_probe.py $sh 0,13,12,19,11 2.0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/energy_r4a.txt; done
for gm in 4 8 16 2; do echo "== IR_LIN_GM=$gm"; IR_LIN_GM=$gm python tools/_lin_ab_sustained.py 8192 3840 1280 0 1 256x256 2>&1 | tail -1; IR_LIN_GM=$gm python tools/_lin_ab_sustained.py 32768 1920 640 0 1 256x256 2>&1 | tail -1; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gm_time.txt
