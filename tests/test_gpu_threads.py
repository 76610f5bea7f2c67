"""Python threads sharing ONE HIP stream (torch's default stream is shared by all threads of a process - a server that handles two
requests in two threads does exactly this).  A call into ``ir_shared_attn_fwd`` issues the attention kernel and, for the K/V-range
pieces of the remainder split, the kernel that merges them through a scratch buffer; ctypes releases the GIL for the call, so two
threads' launches interleave on the stream.  The scratch is therefore per (device, stream, THREAD): with one buffer per stream,
thread B's pieces could overwrite thread A's before A's merge has run (a window of microseconds between the two launches of one
call: not seen failing with the shared buffer, closed on principle).  Every thread's results must equal the single-threaded ones
bit for bit."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_threads_on_the_default_stream_do_not_share_the_split_scratch():
    from instantrestore_amd import ops
    torch.manual_seed(9)
    dt = torch.bfloat16
    QC = 0.125 * 1.4426950408889634
    jobs = []
    for (B, H, L, N) in ((8, 10, 1024, 4), (8, 10, 1024, 3), (3, 10, 1024, 4), (8, 5, 4096, 1)):    # shapes whose last round is split
        C = H * 64
        q = (torch.randn(B, L, C, device="cuda") * QC).to(dt)
        k, v = torch.randn(B, L, C, device="cuda").to(dt), torch.randn(B, L, C, device="cuda").to(dt)
        rk, rv = torch.randn(B, N, L, C, device="cuda").to(dt), torch.randn(B, N, L, C, device="cuda").to(dt)
        aff = ops.adain_stats(v, rv, heads=H)
        kw = dict(heads=H, scale=0.125, include_self=True, adain=aff, q_prescaled=True)
        want = ops.shared_attention(q, k, v, rk, rv, **kw)
        jobs.append(((q, k, v, rk, rv), kw, want))
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        args, kw, want = jobs[i % len(jobs)]
        try:
            for _ in range(25):
                got = ops.shared_attention(*args, **kw)
                if not torch.equal(got, want):
                    errors.append((i, float((got.float() - want.float()).abs().max())))
                    return
        except Exception as e:      # noqa: BLE001 - reported below
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
