mkdir -p gpurun_out/r2g
python -m pytest tests/test_gpu_round2.py tests/test_golden_r2.py -m gpu -x -q 2>&1 | tail -5
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "prescal or processor" 2>&1 | tail -5
for r in 1 2 3; do for v in main EX; do
  if [ $v = main ]; then L=instantrestore_amd/libinstantrestore_hip.so; else L=gpurun_lib/libir_$v.so; fi
  IR_LIB_PATH=$L python tools/_abl_time.py 13 presc
  IR_LIB_PATH=$L python tools/_abl_time.py 13 presc adain
done; done 2>&1 | grep -v amdgpu.ids > gpurun_out/r2g/ab5.txt
sort gpurun_out/r2g/ab5.txt
