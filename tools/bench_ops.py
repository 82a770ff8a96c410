#!/usr/bin/env python
"""Micro-benchmark of the igemm / attention kernels on the layer shapes of the SDv1.5 U-Net at the
bench batch (160 samples, 64x64 latents).  Usage (GPU box): python tools/bench_ops.py [igemm|attn|all]
Variant selection is by env (DM_IGEMM=0..3), latched per process."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import gpu_util as U  # noqa: E402

B = int(os.environ.get("DM_BENCH_B", "160"))
# (name, mode, H, W, C1, C2, Cout, epi)
IGEMM_SHAPES = [
    ("conv3 320->320 @64", 1, 64, 64, 320, 0, 320, 0),
    ("conv3 960->320 @64 (cat)", 1, 64, 64, 640, 320, 320, 0),
    ("conv3 640->640 @32", 1, 32, 32, 640, 0, 640, 0),
    ("conv3 1280->1280 @16", 1, 16, 16, 1280, 0, 1280, 0),
    ("conv3 1280->1280 @8", 1, 8, 8, 1280, 0, 1280, 0),
    ("conv3 2560->1280 @16 (cat)", 1, 16, 16, 1280, 1280, 1280, 0),
    ("geglu 320->2560 @64", 0, 64, 64, 320, 0, 2560, 1),
    ("plain 320->2560 @64", 0, 64, 64, 320, 0, 2560, 0),
    ("plain 320->1280 @64", 0, 64, 64, 320, 0, 1280, 0),
    ("ff2 1280->320 @64", 0, 64, 64, 1280, 0, 320, 0),
    ("qkv 320->960 @64", 0, 64, 64, 320, 0, 960, 0),
    ("proj 320->320 @64", 0, 64, 64, 320, 0, 320, 0),
    ("geglu 640->5120 @32", 0, 32, 32, 640, 0, 5120, 1),
    ("ff2 2560->640 @32", 0, 32, 32, 2560, 0, 640, 0),
    ("down s2 320->320 @64->32", 2, 64, 64, 320, 0, 320, 0),
    ("up 640->640 @32->64", 3, 32, 32, 640, 0, 640, 0),
]
ATTN_SHAPES = [("self D40 T4096", 40, 4096, 4096), ("self D80 T1024", 80, 1024, 1024), ("self D160 T256", 160, 256, 256),
               ("cross D40 T4096x77", 40, 4096, 77), ("cross D80 T1024x77", 80, 1024, 77)]


def timeit(fn, iters=int(os.environ.get("DM_BENCH_ITERS", "5"))):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def bench_igemm():
    lib = U.E.load_library()
    d = U.dev()
    print(f"# igemm variant DM_IGEMM={os.environ.get('DM_IGEMM', 'default')}  B={B}")
    tot_ms = 0
    for name, mode, H, W, C1, C2, Cout, epi in IGEMM_SHAPES:
        taps = 1 if mode == 0 else 9
        x = (torch.randn(B, H, W, C1, device=d) * 0.5).half()
        x2 = (torch.randn(B, H, W, C2, device=d) * 0.5).half() if C2 else None
        w = (torch.randn(Cout, taps * (C1 + C2), device=d) * (taps * (C1 + C2)) ** -0.5).half()
        bias = torch.zeros(Cout, device=d).half()
        OH, OW = (H, W) if mode in (0, 1) else ((H // 2, W // 2) if mode == 2 else (2 * H, 2 * W))
        y = torch.empty(B, OH, OW, Cout // 2 if epi else Cout, device=d, dtype=torch.float16)
        st = U.stream()

        def run():
            rc = lib.dm_op_igemm(st, U.ptr(x), U.ptr(x2), U.ptr(w), U.ptr(bias), None, None, U.ptr(y),
                                 B, H, W, C1, C2, Cout, OH, OW, mode, epi, 0)
            assert rc == 0
        ms = timeit(run)
        flops = 2.0 * B * OH * OW * Cout * taps * (C1 + C2)
        tot_ms += ms
        print(f"{name:32s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TF/s")
    print(f"sum {tot_ms:.2f} ms")


def bench_attn():
    lib = U.E.load_library()
    d = U.dev()
    heads = 8
    print(f"# attention DM_ATTN={os.environ.get('DM_ATTN', 'default')} B={B}")
    for name, D, Tq, Tk in ATTN_SHAPES:
        C = heads * D
        if Tk == Tq:
            qkv = torch.randn(B, Tq, 3 * C, device=d).half()
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
            slots = None
        else:
            q = torch.randn(B, Tq, C, device=d).half()
            kv = torch.randn(2, Tk, 2 * C, device=d).half()
            k, v = kv[..., :C], kv[..., C:]
            slots = (torch.arange(B, device=d) % 2).int()
        o = torch.empty(B, Tq, C, device=d, dtype=torch.float16)
        st = U.stream()

        def run():
            rc = lib.dm_op_attention(st, U.ptr(q), U.ptr(k), U.ptr(v), U.ptr(o), q.stride(1), k.stride(1), v.stride(1), C,
                                     q.stride(0), k.stride(0), v.stride(0), Tq * C, U.ptr(slots), B, heads, Tq, Tk, D,
                                     float(D) ** -0.5)
            assert rc == 0
        ms = timeit(run)
        flops = 4.0 * B * heads * Tq * Tk * D
        print(f"{name:24s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TF/s")


def bench_norm():
    d = U.dev()
    for C, HW in ((320, 4096), (640, 1024), (1280, 256)):
        rows = B * HW
        x = torch.randn(rows, C, device=d).half()
        g = torch.ones(C, device=d); b = torch.zeros(C, device=d)
        y = torch.empty_like(x)
        lib = U.E.load_library(); st = U.stream()
        ms = timeit(lambda: lib.dm_op_layernorm(st, U.ptr(x), rows, C, U.ptr(g), U.ptr(b), 1e-5, U.ptr(y)))
        print(f"layernorm C={C} rows={rows}: {ms:.3f} ms  {2 * rows * C * 2 / ms / 1e9:.2f} TB/s")
    # GroupNorm(32) + SiLU: statistics pass (read) + apply pass (read + write) = 3 x the tensor
    lib = U.E.load_library(); st = U.stream()
    for C1, C2, HW in ((320, 0, 4096), (640, 320, 4096), (320, 320, 4096), (640, 0, 1024), (1280, 640, 1024), (1280, 0, 256), (1280, 1280, 256), (1280, 0, 64)):
        C = C1 + C2
        x = torch.randn(B, HW, C1, device=d).half()
        x2 = torch.randn(B, HW, C2, device=d).half() if C2 else None
        g = torch.ones(C, device=d); b = torch.zeros(C, device=d)
        y = torch.empty(B, HW, C, device=d, dtype=torch.float16)
        ms = timeit(lambda: lib.dm_op_groupnorm(st, U.ptr(x), U.ptr(x2), B, HW, C, C1, 32, 1e-5, U.ptr(g), U.ptr(b), 1, U.ptr(y)), iters=10)
        print(f"groupnorm C={C1}+{C2} HW={HW}: {ms * 1e3:8.1f} us  {3 * B * HW * C * 2 / ms / 1e9:.2f} TB/s (2 reads + 1 write)")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("igemm", "all"):
        bench_igemm()
    if what in ("attn", "all"):
        bench_attn()
    if what in ("norm", "all"):
        bench_norm()
