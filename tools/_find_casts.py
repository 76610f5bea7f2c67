"""which ATen ops run inside the bench step besides this library's kernels (casts, copies, cats): torch.profiler with shapes and stacks
usage: _find_casts.py [two_streams 0/1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
two = len(sys.argv) > 1 and sys.argv[1] == "1"
dev = torch.device("cuda", 0)
layers, (B, N, px, dtype, use_adain) = bench.build_workload("cfg2", True, dev, seed=1234)
bench._AUTOCAST["dtype"] = dtype
with torch.no_grad():
    for _ in range(2):
        bench.hot_path_step(layers, B, N, False, two)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        bench.hot_path_step(layers, B, N, False, two)
        torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=60, max_shapes_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=14, max_name_column_width=50, max_src_column_width=110))
