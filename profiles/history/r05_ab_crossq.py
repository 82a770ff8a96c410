#!/usr/bin/env python
"""LN2 -> to_q -> 77-key cross-attention: the fused kernel (attention_crossq.hip) against the two launches it replaces, at the
bench batch (160 samples x 4096 tokens x 320 channels):   python tools/ab_crossq.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import gpu_util as U  # noqa: E402


def main():
    lib = U.E.load_library()
    d = U.dev()
    B, T, heads, D, Tk, P = int(os.environ.get("B", "160")), 4096, 8, 40, 77, 9
    C = heads * D
    x = (torch.randn(B, T, C, device=d) + 0.3).half()
    Wf = (torch.randn(C, C, device=d) * C ** -0.5).half()
    ln_s, ln_t = Wf.float().sum(1).contiguous(), torch.randn(C, device=d) * 0.1
    kv = torch.randn(P, Tk, 2 * C, device=d).half()
    slots = (torch.arange(B, device=d) % P).int()
    q = torch.empty(B, T, C, dtype=torch.float16, device=d)
    o = torch.empty_like(q)
    st = U.stream()

    def chain():
        assert lib.dm_op_igemm_ln(st, U.ptr(x), U.ptr(Wf), U.ptr(ln_s), U.ptr(ln_t), None, U.ptr(q), B * T, C, C, 0) == 0
        assert lib.dm_op_attention(st, U.ptr(q), U.ptr(kv), U.ptr(kv[..., C:]), U.ptr(o), C, 2 * C, 2 * C, C, T * C, Tk * 2 * C, Tk * 2 * C, T * C,
                                   U.ptr(slots), B, heads, T, Tk, D, float(D) ** -0.5) == 0

    def gemm_only():
        assert lib.dm_op_igemm_ln(st, U.ptr(x), U.ptr(Wf), U.ptr(ln_s), U.ptr(ln_t), None, U.ptr(q), B * T, C, C, 0) == 0

    def fused():
        assert lib.dm_op_cross_attention_q(st, U.ptr(x), U.ptr(Wf), U.ptr(ln_s), U.ptr(ln_t), 1e-5, U.ptr(kv), U.ptr(kv[..., C:]), U.ptr(o),
                                           2 * C, 2 * C, Tk * 2 * C, Tk * 2 * C, U.ptr(slots), B, heads, T, Tk, D, float(D) ** -0.5) == 0

    def timeit(f, n=10):
        f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    best = {}
    for _ in range(3):
        for name, f in (("two launches", chain), ("  of which the LayerNorm-folded GEMM", gemm_only), ("fused", fused)):
            best[name] = min(best.get(name, 1e9), timeit(f))
    for k, v in best.items():
        print(f"{k:40s} {v:7.3f} ms")


if __name__ == "__main__":
    main()
