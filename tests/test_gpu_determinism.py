"""The bench step must be the SAME BITS however it is scheduled: one stream (the reference's own
schedule, inference/test.py:79-111), two streams, and the two-stream step replayed from one hipGraph - ten runs each,
every layer output and every harvested K/V tensor compared with `torch.equal` against the first one-stream run.
Since round 3 every kernel in the step is this library's (no vendor GEMM, no atomics, no memory-side reductions whose
order could move), so a mismatch here is an ordering bug between the streams or an uninitialised read - never noise."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _report(cfg, train_input, reps, modes=("one_stream", "two_streams", "hip_graph")):
    import bench
    dev = torch.device("cuda", 0)
    layers, (B, N, px, dtype, use_adain) = bench.build_workload(cfg, train_input, dev, seed=1234)
    saved = bench._AUTOCAST["dtype"]
    bench._AUTOCAST["dtype"] = dtype
    try:
        with torch.no_grad():
            for _ in range(2):
                bench.hot_path_step(layers, B, N, False, True)
            torch.cuda.synchronize()
            return bench.determinism_report(layers, B, N, reps=reps, modes=modes)
    finally:
        bench._AUTOCAST["dtype"] = saved
        del layers
        torch.cuda.empty_cache()


def _fail_message(rep):
    """which tensor of which layer differed in which mode, and where: what is needed to bisect (tools/gpu_determinism_soak.py)"""
    lines = []
    for mode, r in rep.items():
        for d in r["details"]:
            lines.append("%s rep %d: %s (layer %d: L=%d C=%d H=%d) %d of %d elements differ, first at %s: got %r expected %r, max|diff| %.3e"
                         % (mode, d["repetition"], d["tensor"], d["layer"], d["L"], d["C"], d["H"], d["elements"], d["of"],
                            d["first_index"], d["got"], d["expected"], d["max_abs"]))
    return "\n".join(lines)


@pytest.mark.parametrize("cfg,train_input", [("cfg2", True), ("cfg2", False), ("cfg4", True)])
def test_bench_step_is_bit_identical_on_one_stream_two_streams_and_graph_replay(cfg, train_input):
    """STRICT (round 5): any mismatch in any of the 3 x 10 runs fails, with the tensor, layer, mode and first differing
    element in the message.  Round 4 let a non-reproducing mismatch pass as xfail; an intermittent ordering bug is exactly
    the thing that does not reproduce on demand."""
    rep = _report(cfg, train_input, reps=10)
    assert all(r["identical"] for r in rep.values()), _fail_message(rep)


def test_soak_two_streams_and_graph_replay_cfg4():
    """>= 200 repetitions of the modes in which kernels of the two UNets share the chip (cfg 4: the configuration of the one
    mismatch ever seen, round 4's first box), so that a 1-in-100 ordering bug shows up in the driver's own run"""
    rep = _report("cfg4", True, reps=100, modes=("two_streams", "hip_graph"))
    assert all(r["identical"] for r in rep.values()), _fail_message(rep)
    rep = _report("cfg4", False, reps=100, modes=("two_streams", "hip_graph"))
    assert all(r["identical"] for r in rep.values()), _fail_message(rep)


def test_no_vendor_gemm_in_the_step():
    """every projection of the step resolves to one of this library's kernels (the processors call F.linear otherwise)"""
    import bench
    from instantrestore_amd import ops
    for (B, N, px, dt, _) in bench.CONFIGS.values():
        from instantrestore_amd.roofline import layer_classes
        for (L, C, H) in layer_classes(px):
            for rows in (B * N * L, B * L):
                assert ops.linear_kernel_for(rows, 3 * C, C, False) >= 1
                assert ops.linear_kernel_for(rows, C, C, True) >= 1
                # the f-1 cross-attention layers (round 4): q over the image tokens, k and v as ONE GEMM over the 77 text
                # states of width 1024 (attn_processors._project_qkv)
                assert ops.linear_kernel_for(rows, C, C, False) >= 1
            for sets in (B * N, B):
                assert ops.linear_kernel_for(sets * 77, 2 * C, 1024, False) >= 1
