#!/usr/bin/env python3
"""bench.py - throughput of the InstantRestore hot path on MI355X.

One "step" = one pass of the hot path over one batch of
synthetic inputs already resident in HBM:

  1. K/V capture (a-3): the nine decoder self-attention layers of the frozen reference UNet over
     the B*N reference token sets - ``AttnProcessor`` = to_q/k/v + fused plain attention + to_out,
     stashing K/V;
  2. harvest (a-4): re-view the stashes as (B,N,L,C) and zero-fill invalid references;
  3. shared attention (a-1, a-2): the nine decoder layers of the main UNet over the B degraded
     images - ``SharedAttnProcessor`` = to_q/k/v + AdaIN statistics + fused extended attention
     (AdaIN folded) + to_out.

Conv/ResNet/VAE stages of the UNets are stock MIOpen/hipBLASLt work outside the path
(OUT OF SCOPE) and are NOT in the step; ``config.workload`` says so.
``value`` = identities (restored images) per second over all ranks = B_total / step time.

Contract: ``python bench.py --gpus N --steps K --warmup W``; for N > 1 launched by
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``; one rank per GPU,
identities sharded across ranks (no data-path collective: they are independent), barrier +
synchronize around EXACTLY K timed steps, MAX over ranks, one JSON line from rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch
import torch.distributed as dist

CONFIGS = {
    # name: (identities per GPU, refs, px, dtype, use_adain)
    "cfg2": (8, 4, 512, "bf16", True),    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "cfg4": (8, 8, 512, "bf16", True),    # 8 references
    "cfg5": (16, 4, 1024, "f16", True),   # 1024 px
}
DT = {"bf16": torch.bfloat16, "f16": torch.float16}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_workload(cfg, train_input, dev, seed):
    from face_replace.models.attn_processors import AttnProcessor, SharedAttnProcessor
    from instantrestore_amd.attention import Attention
    from instantrestore_amd.roofline import layer_classes

    B, N, px, dt, use_adain = CONFIGS[cfg]
    dtype = DT[dt]
    g = torch.Generator(device="cpu").manual_seed(seed)
    layers = []
    idx = 0
    for (L, C, H) in layer_classes(px):
        for _ in range(3):
            def mk(proc):
                a = Attention(query_dim=C, heads=H, dim_head=64, processor=proc)
                with torch.no_grad():
                    for lin in (a.to_q, a.to_k, a.to_v, a.to_out[0]):
                        lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) / C ** 0.5)
                return a.to(dev, dtype)
            kv_attn = mk(AttnProcessor())
            main_attn = mk(SharedAttnProcessor(self_attn_idx=idx, use_adain=use_adain, train_input=train_input))
            # token activations entering the two attentions (LayerNorm output in the real UNet)
            h_ref = torch.randn(B * N, L, C, generator=g).to(dev, dtype)
            h_main = torch.randn(B, L, C, generator=g).to(dev, dtype)
            layers.append(dict(L=L, C=C, H=H, kv_attn=kv_attn, main_attn=main_attn, h_ref=h_ref, h_main=h_main))
            idx += 1
    return layers, (B, N, px, dtype, use_adain)


_REF_STREAM = {}


def hot_path_step(layers, B, N, ref_early_exit=False, two_streams=False):
    """one pass: K/V capture -> harvest -> shared attention.  Returns the 9 outputs.

    ``two_streams``: the reference UNet's layers run on their own HIP stream and shared layer i waits only
    for the event of capture layer i (it reads nothing else), so the small layer classes - which cannot
    fill 256 CUs on their own - overlap with the other UNet's kernels.  Same kernels, same work."""
    from instantrestore_amd.attn_processors import ReferenceCaptureComplete
    cur = torch.cuda.current_stream()
    ref_stream = cur
    if two_streams:
        ref_stream = _REF_STREAM.setdefault(cur.device, torch.cuda.Stream(device=cur.device))
        ref_stream.wait_stream(cur)
    # 1. K/V capture on the reference token sets (--ref-early-exit: the reference UNet stops after to_k / to_v
    #    of its last capturing layer, SURVEY 8f rank 2 - NOT the default, the headline runs all nine in full)
    procs = [ly["kv_attn"].processor for ly in layers]
    for p in procs:
        p.stop_after_capture = procs if ref_early_exit else None
    for p in procs:
        p.record_events = two_streams
    with torch.cuda.stream(ref_stream):
        for ly in layers:
            try:
                ly["kv_attn"](ly["h_ref"])
            except ReferenceCaptureComplete:
                pass
    # 2. harvest (views) + zero-fill of invalid references (valid = N at inference, test.py:81)
    keys, vals, events = [], [], []
    for ly in layers:
        p = ly["kv_attn"].processor
        keys.append(p.keys.reshape(-1, N, p.keys.shape[1], p.keys.shape[2]))
        vals.append(p.values.reshape(-1, N, p.values.shape[1], p.values.shape[2]))
        events.append(p.ready)
        p.reset()
    # 3. shared attention on the degraded images (each layer waits for its own reference layer's event)
    outs = []
    for ly in layers:
        outs.append(ly["main_attn"](ly["h_main"], ref_keys=keys, ref_values=vals,
                                    ref_events=events if two_streams else None))
    if two_streams:
        cur.wait_stream(ref_stream)
    return outs


def measure_roofline(layers, B, N, train_input, use_adain, dtype):
    """dominant kernel = fused shared attention of the 64x64-token layer class; timed live with
    HIP events on the launch stream (ir_time_shared_attn_fwd)."""
    from instantrestore_amd import ops
    from instantrestore_amd.roofline import MFMA_PEAK_TFLOPS_16BIT, attn_bytes, attn_flops

    ly = layers[-1]
    L, C, H = ly["L"], ly["C"], ly["H"]
    a = ly["main_attn"]
    with torch.no_grad():
        q, k, v = a.to_q(ly["h_main"]), a.to_k(ly["h_main"]), a.to_v(ly["h_main"])
        kr = layers[-1]["kv_attn"].to_k(ly["h_ref"]).reshape(B, N, L, C)
        vr = layers[-1]["kv_attn"].to_v(ly["h_ref"]).reshape(B, N, L, C)
        aff = ops.adain_stats(v, vr, heads=H) if use_adain else None
        ops.time_shared_attention(q, k, v, kr, vr, heads=H, scale=0.125, include_self=train_input, adain=aff, iters=3)
        ms = ops.time_shared_attention(q, k, v, kr, vr, heads=H, scale=0.125, include_self=train_input,
                                       adain=aff, iters=20)
        kname = ops.shared_attention_kernel_name(q, k, v, kr, vr, heads=H, scale=0.125, include_self=train_input, adain=aff)
    lkv = (N + int(train_input)) * L
    flops = attn_flops(B, L, lkv, C)
    tf = flops / (ms * 1e-3) / 1e12
    return {
        "bound": "mfma",
        "kernel": "%s %s (L=%d, Lkv=%d, H=%d, B=%d)" % (
            kname, {torch.bfloat16: "bf16", torch.float16: "f16"}[dtype], L, lkv, H, B),
        "achieved": round(tf, 2),
        "peak": MFMA_PEAK_TFLOPS_16BIT,
        "unit": "TFLOP/s",
        "frac": round(tf / MFMA_PEAK_TFLOPS_16BIT, 4),
        "ms_per_launch": round(ms, 4),
        "algorithmic_gflop_per_launch": round(flops / 1e9, 2),
        "algorithmic_mb_per_launch": round(attn_bytes(B, L, lkv, C) / 1e6, 2),
        # HBM bytes per launch from a separate rocprofv3 --pmc pass of this kernel at this shape
        # (tools/pmc_attn.sh -> profiles/r1_pmc_shared_attn.txt); FETCH_SIZE doubled per the
        # gfx950 correction of MI355X_MICROARCH.md.  null if the profile is not for this shape.
        "traffic": _pmc_traffic_bytes() if (B, N, L, H, train_input, use_adain) == (8, 4, 4096, 5, True, True) else None,
        "traffic_unit": "bytes/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, profiles/r1_pmc_shared_attn.txt)",
        "power_note": "this kernel runs at the 1400 W board cap on random data (profiles/r1_power_probe.txt): "
                      "sustained shader clock 2.1-2.2 GHz against the 2.4 GHz the peak assumes; an MFMA-only stream of the same "
                      "instruction holds 1.96 PFLOP/s on random operands under that cap (profiles/r1_ubench_mfma_power.txt)",
    }


def _pmc_traffic_bytes():
    import ast
    path = os.path.join(REPO, "profiles", "r1_pmc_shared_attn.txt")
    try:
        vals = {}
        for line in open(path):
            i, j = line.find("{"), line.rfind("}")
            if i >= 0 and j > i:
                vals.update(ast.literal_eval(line[i:j + 1]))
        return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
    except Exception:
        return None


def cpu_baseline(N, px, train_input, use_adain, seed, budget_s=25.0):
    """oracle port (torch-CPU fp32, the reference's operator sequence) on the host cores: one
    identity through the same 9+9 layers, bounded to ~budget_s of CPU work."""
    from instantrestore_amd.roofline import layer_classes
    from oracle import shared_attn_oracle as O

    torch.manual_seed(seed)
    cores = torch.get_num_threads()
    t_total, done_layers, n_layers = 0.0, 0, 0
    per_class = []
    for (L, C, H) in layer_classes(px):
        n_layers += 3
        if t_total > budget_s:
            per_class.append(None)
            continue
        w = [torch.randn(C, C) / C ** 0.5 for _ in range(4)]
        bo = torch.zeros(C)
        h_ref, h_main = torch.randn(N, L, C), torch.randn(1, L, C)
        t0 = time.perf_counter()
        # K/V capture: plain attention over the N reference token sets
        O.shared_attn_processor_port(h_ref, w[0], w[1], w[2], w[3], bo, None, None, H)
        kr = torch.nn.functional.linear(h_ref, w[1]).reshape(1, N, L, C)
        vr = torch.nn.functional.linear(h_ref, w[2]).reshape(1, N, L, C)
        O.shared_attn_processor_port(h_main, w[0], w[1], w[2], w[3], bo, kr, vr, H, use_adain, train_input)
        dt = time.perf_counter() - t0
        per_class.append(dt)
        t_total += 3 * dt  # three identical layers per class: time one, count three
        done_layers += 3
    complete = done_layers == n_layers
    value = (1.0 / t_total) if complete and t_total > 0 else None
    return {
        "value": None if value is None else round(value, 4),
        "unit": "images/s",
        "cores": cores,
        "kind": "port",
        "sample": "1 identity, %d refs, %d px, fp32 torch-CPU port of the reference operator sequence "
                  "(oracle/shared_attn_oracle.py), one layer per class timed and counted x3; seconds per class: %s"
                  % (N, px, [None if t is None else round(t, 3) for t in per_class]),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--train-input", type=int, default=1, help="1: self K/V precede the reference K/V (t=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant for A/B (ir_set_attn_variant); 0 = default")
    ap.add_argument("--two-streams", type=int, default=1,
                    help="1: reference-UNet layers on their own HIP stream, shared layer i waits for capture layer i only")
    ap.add_argument("--ref-early-exit", action="store_true",
                    help="stop the reference UNet after the K/V projections of its last capturing layer (its output is "
                         "discarded by the inference caller); off for the headline number")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    # test hooks (tests/test_bench_contract.py drives the N = 2 control flow on a one-GPU box): every rank on
    # cuda:0 and the two control-plane collectives (barrier, MAX of the elapsed time) over gloo
    backend = os.environ.get("IR_BENCH_DIST_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("IR_BENCH_SHARE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    train_input = bool(args.train_input)
    if args.variant:
        from instantrestore_amd import ops as _ops
        _ops.set_attn_variant(args.variant)
    layers, (B, N, px, dtype, use_adain) = build_workload(args.config, train_input, dev, seed=1234 + rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            outs = hot_path_step(layers, B, N, args.ref_early_exit, bool(args.two_streams))
        barrier()
        # HIP events on the launch stream around the dominant kernel's launches INSIDE the timed region
        # (the shared attention of the largest layer class: 3 launches per step)
        from instantrestore_amd import ops as _ops_mod
        top_l = layers[-1]["L"]
        in_step = []
        if rank == 0 and not args.no_roofline:
            _ops_mod.EVENT_SINK = (lambda q, rk, ad: rk is not None and q.shape[1] == top_l, in_step)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            outs = hot_path_step(layers, B, N, args.ref_early_exit, bool(args.two_streams))
        barrier()   # synchronize + barrier + synchronize: the K steps are bracketed on both sides
        elapsed = time.perf_counter() - t0
        _ops_mod.EVENT_SINK = None
        in_step_ms = [a.elapsed_time(b) for a, b in in_step]
    assert all(torch.isfinite(o).all() for o in outs)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    roof = cpu = None
    if rank == 0:
        if not args.no_roofline:
            # `achieved`: the dominant kernel's average launch duration INSIDE timed steps, HIP events on its
            # launch stream - over a second timed region of the same K steps run on ONE stream, because with two
            # streams the reference UNet's kernels share the chip with it and the in-situ time stops being the
            # kernel's own.  The back-to-back figure (20 launches on an idle GPU, measured afterwards) is kept
            # next to it: a sustained run of this one kernel clocks lower than the same kernel between the
            # step's lighter kernels.
            one_stream = []
            _ops_mod.EVENT_SINK = (lambda q, rk, ad: rk is not None and q.shape[1] == top_l, one_stream)
            with torch.no_grad():
                for _ in range(args.steps):
                    hot_path_step(layers, B, N, args.ref_early_exit, False)
            torch.cuda.synchronize()
            _ops_mod.EVENT_SINK = None
            roof = measure_roofline(layers, B, N, train_input, use_adain, dtype)
            b2b = {k: roof[k] for k in ("achieved", "frac", "ms_per_launch")}
            ms_in = sum(a.elapsed_time(b) for a, b in one_stream) / len(one_stream)
            tf_in = roof["algorithmic_gflop_per_launch"] / ms_in
            roof.update({"achieved": round(tf_in, 2), "frac": round(tf_in / roof["peak"], 4), "ms_per_launch": round(ms_in, 4),
                         "launches_timed": len(one_stream),
                         "timing": "HIP events on the launch stream around every launch of this kernel inside K timed steps (one stream)",
                         "back_to_back": b2b})
            if in_step_ms and args.two_streams:
                ms2 = sum(in_step_ms) / len(in_step_ms)
                roof["in_step_two_streams"] = {"ms_per_launch": round(ms2, 4), "launches": len(in_step_ms),
                                               "note": "in the headline run: includes time shared with the other stream's kernels"}
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(N, px, train_input, use_adain, seed=99)
    if world > 1:
        dist.barrier()

    if rank == 0:
        from instantrestore_amd.roofline import summary
        ms = elapsed / args.steps * 1e3
        total_ids = B * world
        line = {
            "metric": "restored images/sec @512px, 4 refs, single-step; 1/2/4/8 MI355X",
            "value": round(total_ids / (elapsed / args.steps), 3),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {torch.bfloat16: "bf16", torch.float16: "f16"}[dtype],
            "data": "synthetic",
            "config": {
                "workload": "%s: hot path only (SURVEY 8a a-1..a-4) = 9 K/V-capture layers over B*N reference token "
                            "sets + harvest + 9 shared-attention layers (to_q/k/v, AdaIN stats, fused extended "
                            "attention, to_out) over B identities; UNet conv/ResNet and VAE stages are out of scope "
                            "and not in the step" % args.config,
                "identities_per_gpu": B, "global_batch": total_ids, "refs": N, "px": px,
                "use_adain": use_adain, "train_input": train_input, "ref_early_exit": bool(args.ref_early_exit), "two_streams": bool(args.two_streams), "parallelism": "dp%d (independent identities)" % world,
                **{k: round(v, 1) for k, v in summary(N, train_input, px).items()},
            },
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
