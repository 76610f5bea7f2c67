"""ir_bench_mfma_stream (ABI v7): the MFMA-only stream bench.py uses for `roofline.at_power_cap`.  Measurement aid, so the
checks are sanity bounds: the result is a plausible matrix-pipe rate for gfx950 (dense bf16/fp16 peak 2.5 PFLOP/s at
2.4 GHz), all-zero operands are at least as fast as pseudo-random ones (no data-dependent switching: the board stays off
its power cap), (argument validation: tests/test_cabi_symbols.py, no GPU needed)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_mfma_stream_rates_are_plausible(dtype):
    from instantrestore_amd import ops
    rnd = ops.bench_mfma_stream(dtype, zero_operands=False, iters=20000, launches=3)
    zer = ops.bench_mfma_stream(dtype, zero_operands=True, iters=20000, launches=3)
    assert 800.0 < rnd < 2700.0 and 800.0 < zer < 2700.0, (rnd, zer)
    assert zer >= 0.97 * rnd, (rnd, zer)
