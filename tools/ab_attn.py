#!/usr/bin/env python
"""A/B of a runtime switch on the U-Net's attention shapes at the bench batch (both arms in one process, interleaved):
    python tools/ab_attn.py attn_pipe 0 1"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import gpu_util as U  # noqa: E402

B0, HEADS = 160, 8
# (name, n/step, D, Tq, Tk)   Tk = 77: cross-attention against 2 prompts
SHAPES = [("cross D40 T4096", 5, 40, 4096, 77), ("cross D80 T1024", 5, 80, 1024, 77), ("cross D160 T256", 5, 160, 256, 77),
          ("self D40 T4096", 5, 40, 4096, 4096), ("self D80 T1024", 5, 80, 1024, 1024), ("self D160 T256", 5, 160, 256, 256), ("self D80 T4096 (x-ray, batch 20)", 5, 80, 4096, 4096)]


def main():
    opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    iters = int(os.environ.get("DM_BENCH_ITERS", "5"))
    lib = U.E.load_library()
    d = U.dev()
    print(f"# {opt}: A = {va}, B = {vb}; batch {B0}; ms per launch (min of 3 interleaved rounds of {iters})")
    tot = [0.0, 0.0]
    for name, n, D, Tq, Tk in SHAPES:
        B = 20 if "batch 20" in name else B0
        C = HEADS * D
        cross = Tk == 77
        q = torch.randn(B, Tq, C, device=d).half()
        if cross:
            kv = torch.randn(2, Tk, 2 * C, device=d).half()
            k, v = kv[..., :C], kv[..., C:]
            slots = (torch.arange(B, device=d) // (B // 2)).int()
        else:
            kv = torch.randn(B, Tk, 2 * C, device=d).half()
            k, v = kv[..., :C], kv[..., C:]
            slots = None

        def timeit():
            U.op_attention(q, k, v, HEADS, slots=slots)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                U.op_attention(q, k, v, HEADS, slots=slots)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / iters
        best = [1e9, 1e9]
        for _ in range(3):
            for i, val in enumerate((va, vb)):
                assert lib.dm_set_option(opt.encode(), val) == 0
                best[i] = min(best[i], timeit())
        tot[0] += n * best[0]
        tot[1] += n * best[1]
        print(f"{name:20s} x{n}  A {best[0]:7.3f} ms   B {best[1]:7.3f} ms   B/A {best[1] / best[0]:.3f}", flush=True)
    print(f"per step (these shapes): A {tot[0]:.2f} ms  B {tot[1]:.2f} ms")


if __name__ == "__main__":
    main()
