mkdir -p gpurun_out/r2i
python -m pytest tests/test_gpu_linear.py -m gpu -x -q 2>&1 | tail -3
IR_LIB_PATH=gpurun_lib/libir_N8.so python -m pytest tests/test_gpu_linear.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do for v in N4 N8; do
  IR_LIB_PATH=gpurun_lib/libir_$v.so python tools/_lin_time.py 2>&1 | grep "K=320"
done; done 2>&1 | grep -v amdgpu.ids > gpurun_out/r2i/lin4.txt
sort gpurun_out/r2i/lin4.txt
