#!/usr/bin/env python
"""A/B of the LayerNorm statistics kernels (`ln_stats_g` 0 / 1) on the U-Net's token matrices at the bench batch, both arms
interleaved in one process:   python tools/ab_ln_stats.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import gpu_util as U  # noqa: E402

lib = U.E.load_library()
d = U.dev()
tot = [0.0, 0.0]
for name, n, rows, C in [("64x64 C=320", 15, 160 * 4096, 320), ("32x32 C=640", 15, 160 * 1024, 640), ("16x16 C=1280", 15, 160 * 256, 1280),
                         ("8x8 C=1280", 3, 160 * 64, 1280)]:
    x = torch.randn(rows, C, device=d).half()
    stats = torch.empty(rows, 2, dtype=torch.float32, device=d)
    st = U.stream()

    def timeit(iters=20):
        for _ in range(2):
            assert lib.dm_op_ln_stats(st, U.ptr(x), rows, C, 1e-5, U.ptr(stats)) == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            lib.dm_op_ln_stats(st, U.ptr(x), rows, C, 1e-5, U.ptr(stats))
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters
    best = [1e9, 1e9]
    for _ in range(3):
        for k, v in enumerate((0, 1)):
            assert lib.dm_set_option(b"ln_stats_g", v) == 0
            best[k] = min(best[k], timeit())
    gb = rows * C * 2 / 1e9
    tot[0] += n * best[0]
    tot[1] += n * best[1]
    print(f"{name:14s} x{n}  A {best[0] * 1e3:7.1f} us {gb / best[0]:6.2f} TB/s   B {best[1] * 1e3:7.1f} us {gb / best[1]:6.2f} TB/s   B/A {best[1] / best[0]:.3f}")
print(f"per step: A {tot[0]:.2f} ms  B {tot[1]:.2f} ms   (back-to-back launches on an L2 / Infinity-Cache-warm tensor for the small ones)")
