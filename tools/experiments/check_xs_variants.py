"""development-build check of the two experimental schedules of the K = 320 X-stationary GEMM (tools/experiments/linear_xs_pp.hip,
linear_xs_rot.hip; ir_linear_fwd_ex kernel ids 9 / 10): the same products accumulated in the same order as the product kernel, so
every result must agree with it BIT FOR BIT - fp32 activations, the column scale, ragged M, odd chunk counts, single chunks,
strided operands and an output that is a column slice of a wider buffer included.
usage: tools/experiments/build.sh && IR_LIB_PATH=tools/experiments/libinstantrestore_hip_dev.so python tools/experiments/check_xs_variants.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from instantrestore_amd import ops, _lib

K = 320
n_checked = 0
for variant, kid in ops.LIN_KERNELS_DEV.items():
    for dtype in (torch.float16, torch.bfloat16):
        for M, N, bias in [(131072, 960, False), (131072, 320, True), (65536, 640, False), (512, 64, True), (1, 32, False), (513, 96, True),
                           (70001, 352, True), (4099, 2880, False), (40000, 32, True)]:
            g = torch.Generator().manual_seed(M + N)
            x32 = torch.randn(M, K, generator=g).cuda()
            w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).cuda()
            b = torch.randn(N, generator=g).to(dtype).cuda() if bias else None
            for x in (x32.to(dtype), x32):
                for kw in ({}, {"scale_cols": 32, "col_scale": 0.125 * 1.4426950408889634}):
                    a = ops.linear(x, w, b, kernel=ops.LIN_KERNELS["x_stationary"], **kw)
                    c = ops.linear(x, w, b, kernel=kid, **kw)
                    assert torch.equal(a, c), (variant, dtype, M, N, bias, x.dtype, kw, (a != c).nonzero()[:4])
                    n_checked += 1
    # integer operands (exact products), strided x rows, a row slice of a fused weight, Y a column slice of a wider buffer
    g = torch.Generator().manual_seed(5)
    M, N = 1500, 160
    xbig = torch.randint(-3, 4, (M, K + 64), generator=g).to(torch.bfloat16).cuda()
    x = xbig[:, :K]
    wbig = torch.randint(-2, 3, (3 * N, K), generator=g).to(torch.bfloat16).cuda()
    b = torch.randint(-4, 5, (N,), generator=g).to(torch.bfloat16).cuda()
    ybig = torch.full((M + 7, N + 64), 7.0, dtype=torch.bfloat16, device="cuda")
    rc = _lib.lib().ir_linear_fwd_ex(1, 0, M, N, K, x.data_ptr(), x.stride(0), wbig[N:2 * N].data_ptr(), K, b.data_ptr(),
                                     ybig.data_ptr(), ybig.stride(0), 0, 1.0, kid, ops._stream())
    _lib.check(rc, "ir_linear_fwd_ex")
    torch.cuda.synchronize()
    ref = (x.float() @ wbig[N:2 * N].float().T + b.float()).to(torch.bfloat16)
    assert torch.equal(ybig[:M, :N], ref)
    assert bool((ybig[:, N:] == 7.0).all()) and bool((ybig[M:] == 7.0).all())
    print(variant, "ok")
print("bit-identical to the product kernel in %d comparisons" % n_checked)
