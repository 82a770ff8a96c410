#!/usr/bin/env python
"""Phase timers of attention_pp.hip (variants 14 = two sets, 15 = three sets): shader cycles of one wave per set, per key tile.
    python tools/attn_pp_timing.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import gpu_util as U  # noqa: E402

NAMES = ["M work", "M barrier", "V1 work", "V1 barrier", "V2 work", "V2 barrier", "tail", "prologue"]


def main():
    lib = U.E.load_library()
    d = U.dev()
    B, T, H, D = 160, 4096, 8, 40
    qkv = torch.randn(B, T, 3 * H * D, device=d).half()
    q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
    for var, nsets in ((14, 2), (15, 3)):
        assert lib.dm_set_option(b"attn_pipe", var) == 0
        for _ in range(2):
            U.op_attention(q, k, v, H)
        out = (C.c_longlong * 24)()
        assert lib.dm_debug_attn_pp_timing(out) == 0
        print(f"variant {var} ({nsets} sets), cycles per key tile ({T // 64} tiles):")
        for s in range(nsets):
            row = [out[s * 8 + i] for i in range(8)]
            per = [r / (T // 64) for r in row[:6]]
            print(f"  set {s}: " + "  ".join(f"{n} {p:7.1f}" for n, p in zip(NAMES[:6], per)) + f"   sum {sum(per):7.1f}   tail {row[6]}  prologue {row[7]}")
    lib.dm_set_option(b"attn_pipe", 1)


if __name__ == "__main__":
    main()
