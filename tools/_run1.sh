mkdir -p gpurun_out/r2i
for v in main LS LL LLS; do
  if [ $v = main ]; then L=instantrestore_amd/libinstantrestore_hip.so; else L=gpurun_lib/libir_$v.so; fi
  IR_LIB_PATH=$L python tools/_lin_time.py 2>&1 | grep "K=320"
done 2>&1 | grep -v amdgpu.ids > gpurun_out/r2i/lin3.txt
cat gpurun_out/r2i/lin3.txt
