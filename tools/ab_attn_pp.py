#!/usr/bin/env python
"""Same-process interleaved timing of the head_dim-40 self-attention variants (`attn_pipe` values) at the two shapes that matter:
4096 tokens x 160 samples (configs[1]) and 16384 tokens x 20 samples (configs[4]):
    python tools/ab_attn_pp.py 1 4 5 6 7 8"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import gpu_util as U  # noqa: E402

HEADS, D = 8, 40
SHAPES = [("self D40 T4096 B160", 160, 4096), ("self D40 T16384 B20", 20, 16384)]


def main():
    variants = [int(a) for a in sys.argv[1:]] or [1, 4]
    iters = int(os.environ.get("DM_BENCH_ITERS", "5"))
    lib = U.E.load_library()
    d = U.dev()
    C = HEADS * D
    for name, B, T in SHAPES:
        qkv = torch.randn(B, T, 3 * C, device=d).half()
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        flops = 4.0 * B * HEADS * T * T * D
        best = {}
        ref = None
        for rnd in range(3):
            for val in variants:
                assert lib.dm_set_option(b"attn_pipe", val) == 0
                o = U.op_attention(q, k, v, HEADS)
                if ref is None:
                    ref = o
                elif rnd == 0:
                    err = (o.float() - ref.float()).norm().item() / ref.float().norm().item()
                    print(f"   variant {val} vs variant {variants[0]}: rel-L2 {err:.2e}", flush=True)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(iters):
                    U.op_attention(q, k, v, HEADS)
                b.record()
                torch.cuda.synchronize()
                best[val] = min(best.get(val, 1e9), a.elapsed_time(b) / iters)
        for val in variants:
            print(f"{name:22s} attn_pipe={val}: {best[val]:7.3f} ms  {flops / best[val] * 1e-9:7.1f} TFLOP/s   x{best[val] / best[variants[0]]:.3f}", flush=True)
    lib.dm_set_option(b"attn_pipe", 1)


if __name__ == "__main__":
    main()
