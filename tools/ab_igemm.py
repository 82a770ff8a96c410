#!/usr/bin/env python
"""A/B of a runtime switch (dm_set_option) on the U-Net's igemm shapes at the bench batch, both arms in ONE process on ONE
box, interleaved (box-to-box and run-to-run spread is +-2-3 %, more than most kernel changes):

    python tools/ab_igemm.py igemm_big 0 1 [substr]       # option, value A, value B (here: 128-row tile vs persistent 256 x 320);
                                                          # substr: only the shapes whose name contains it ("@16")
Shapes: every (mode, M, N, K, epilogue) of a bench step that takes the 256 x 320 tile, with its launches per step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import gpu_util as U  # noqa: E402

B = int(os.environ.get("DM_AB_BATCH", "160"))
# (name, n/step, mode, H, W, C1, C2, Cout, epi, extra)   extra: "" | "temb" | "res" | "ln"
SHAPES = [
    ("ff1 geglu 320->2560 @64 ln", 5, 0, 64, 64, 320, 0, 2560, 1, "ln"),
    ("ff1 geglu 640->5120 @32 ln", 5, 0, 32, 32, 640, 0, 5120, 1, "ln"),
    ("ff1 geglu 1280->10240 @16 ln", 5, 0, 16, 16, 1280, 0, 10240, 1, "ln"),
    ("conv1 3x3 640->640 @32 temb", 3, 1, 32, 32, 640, 0, 640, 0, "temb"),
    ("conv2 3x3 640->640 @32 res", 3, 1, 32, 32, 640, 0, 640, 0, "res"),
    ("conv1 3x3 1280->1280 @16 temb", 3, 1, 16, 16, 1280, 0, 1280, 0, "temb"),
    ("conv2 3x3 1280->1280 @16 res", 3, 1, 16, 16, 1280, 0, 1280, 0, "res"),
    ("conv1 3x3 cat 1280+1280->1280 @16", 2, 1, 16, 16, 1280, 1280, 1280, 0, "temb"),
    ("conv1 3x3 cat 640+320->320 @64", 1, 1, 64, 64, 640, 320, 320, 0, "temb"),
    ("conv1 3x3 cat 320+320->320 @64", 2, 1, 64, 64, 320, 320, 320, 0, "temb"),
    ("conv1 3x3 cat 1280+640->640 @32", 1, 1, 32, 32, 1280, 640, 640, 0, "temb"),
    ("up 3x3 640->640 @32->64", 1, 3, 32, 32, 640, 0, 640, 0, ""),
    ("up 3x3 1280->1280 @16->32", 1, 3, 16, 16, 1280, 0, 1280, 0, ""),
    ("ff2 1280->320 @64 res", 5, 0, 64, 64, 1280, 0, 320, 0, "res"),
    ("ff2 2560->640 @32 res", 5, 0, 32, 32, 2560, 0, 640, 0, "res"),
    ("qkv 640->1920 @32 ln", 5, 0, 32, 32, 640, 0, 1920, 0, "ln"),
    ("shortcut 1x1 cat 640+320->320 @64", 1, 0, 64, 64, 640, 320, 320, 0, ""),
    # shapes the per-shape rule keeps on the 128-row tile (Cin < 640, or < 1024 big tiles)
    ("conv1 3x3 320->320 @64 temb", 3, 1, 64, 64, 320, 0, 320, 0, "temb"),
    ("conv2 3x3 320->320 @64 res", 3, 1, 64, 64, 320, 0, 320, 0, "res"),
    ("proj 320->320 @64 res", 15, 0, 64, 64, 320, 0, 320, 0, "res"),
    ("qkv 320->960 @64 ln", 5, 0, 64, 64, 320, 0, 960, 0, "ln"),
    ("proj 640->640 @32 res", 20, 0, 32, 32, 640, 0, 640, 0, "res"),
    ("proj 1280->1280 @16 res", 20, 0, 16, 16, 1280, 0, 1280, 0, "res"),
    ("ff2 5120->1280 @16 res", 5, 0, 16, 16, 5120, 0, 1280, 0, "res"),
    ("down 3x3 s2 320->320 @64->32", 1, 2, 64, 64, 320, 0, 320, 0, ""),
    # the rest of the 16x16 level (640 tiles of 256 x 320 at the bench batch = 2.5 rounds: "igemm_tail")
    ("conv1 3x3 cat 1280+640->1280 @16", 1, 1, 16, 16, 1280, 640, 1280, 0, "temb"),
    ("conv1 3x3 640->1280 @16 temb", 1, 1, 16, 16, 640, 0, 1280, 0, "temb"),
    ("qkv 1280->3840 @16 ln", 5, 0, 16, 16, 1280, 0, 3840, 0, "ln"),
    ("q2 1280->1280 @16 ln", 5, 0, 16, 16, 1280, 0, 1280, 0, "ln"),
    ("shortcut 1x1 cat 1280+1280->1280 @16", 2, 0, 16, 16, 1280, 1280, 1280, 0, ""),
    ("shortcut 1x1 640->1280 @16", 1, 0, 16, 16, 640, 0, 1280, 0, ""),
    # r03 merged layers (dm_op_igemm_shortcut): extra = "sc:<C3>:<C4>[:res]" — a second GEMM on cat([X3, X4]) as extra k steps
    ("ff2+proj_out [ff|t2] 1280+320->320 @64", 5, 0, 64, 64, 1280, 0, 320, 0, "sc:320:0:res"),
    ("ff2+proj_out [ff|t2] 2560+640->640 @32", 5, 0, 32, 32, 2560, 0, 640, 0, "sc:640:0:res"),
    ("ff2+proj_out [ff|t2] 5120+1280->1280 @16", 5, 0, 16, 16, 5120, 0, 1280, 0, "sc:1280:0:res"),
    ("conv2+shortcut 3x3 640 + cat 640+320 ->640 @32", 1, 1, 32, 32, 640, 0, 640, 0, "sc:640:320"),
    ("conv2+shortcut 3x3 320 + cat 320+320 ->320 @64", 2, 1, 64, 64, 320, 0, 320, 0, "sc:320:320"),
    ("conv2+shortcut 3x3 1280 + cat 1280+1280 ->1280 @16", 2, 1, 16, 16, 1280, 0, 1280, 0, "sc:1280:1280"),
]


def main():
    opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    iters = int(os.environ.get("DM_BENCH_ITERS", "6"))
    lib = U.E.load_library()
    d = U.dev()
    g = torch.Generator(device="cuda").manual_seed(1)
    print(f"# {opt}: A = {va}, B = {vb}; batch {B}; ms per launch (min of 3 interleaved rounds of {iters})")
    tot = [0.0, 0.0]
    cur_val = [0]
    only = sys.argv[4] if len(sys.argv) > 4 else ""
    for name, n, mode, H, W, C1, C2, Cout, epi, extra in SHAPES:
        if only and only not in name:
            continue
        taps = 1 if mode == 0 else 9
        Cin = C1 + C2
        OH, OW = (H, W) if mode in (0, 1) else ((H // 2, W // 2) if mode == 2 else (2 * H, 2 * W))
        M = B * OH * OW
        x = (torch.randn(B, H, W, C1, device=d, generator=g) * 0.5).half()
        x2 = (torch.randn(B, H, W, C2, device=d, generator=g) * 0.5).half() if C2 else None
        w = (torch.randn(Cout, taps * Cin, device=d, generator=g) * (taps * Cin) ** -0.5).half()
        bias = torch.zeros(Cout, device=d).half()
        temb = torch.randn(B, Cout, device=d, generator=g).half() if extra == "temb" else None
        res = torch.randn(B, OH, OW, Cout, device=d, generator=g).half() if extra == "res" else None
        y = torch.empty(B, OH, OW, Cout // 2 if epi else Cout, device=d, dtype=torch.float16)
        st = U.stream()
        if extra.startswith("sc:"):
            f = extra.split(":")
            C3, C4 = int(f[1]), int(f[2])
            x3 = (torch.randn(B, H, W, C3, device=d, generator=g) * 0.5).half()
            x4 = (torch.randn(B, H, W, C4, device=d, generator=g) * 0.5).half() if C4 else None
            w = (torch.randn(Cout, taps * Cin + C3 + C4, device=d, generator=g) * (taps * Cin + C3 + C4) ** -0.5).half()
            res = torch.randn(B, OH, OW, Cout, device=d, generator=g).half() if f[-1] == "res" else None

            def run():
                assert lib.dm_op_igemm_shortcut(st, U.ptr(x), U.ptr(x3), U.ptr(x4), U.ptr(w), U.ptr(bias), U.ptr(res), U.ptr(y),
                                                B, H, W, Cin, C3, C4, Cout, mode) == 0
        elif extra == "ln":
            stats = torch.empty(M, 2, dtype=torch.float32, device=d)
            assert lib.dm_op_ln_stats(st, U.ptr(x), M, Cin, 1e-5, U.ptr(stats)) == 0
            ln_s, ln_t = w.float().sum(1).contiguous(), torch.zeros(Cout, device=d)

            def run():
                # "ln_inkernel": arm 1 = the GEMM takes the row statistics itself (stats = NULL); arm 0 = statistics kernel + GEMM
                if opt == "ln_inkernel":
                    if cur_val[0]:
                        assert lib.dm_op_igemm_ln(st, U.ptr(x), U.ptr(w), U.ptr(ln_s), U.ptr(ln_t), None, U.ptr(y), M, Cin, Cout, epi) == 0
                        return
                    assert lib.dm_op_ln_stats(st, U.ptr(x), M, Cin, 1e-5, U.ptr(stats)) == 0
                assert lib.dm_op_igemm_ln(st, U.ptr(x), U.ptr(w), U.ptr(ln_s), U.ptr(ln_t), U.ptr(stats), U.ptr(y), M, Cin, Cout, epi) == 0
        else:
            def run():
                assert lib.dm_op_igemm(st, U.ptr(x), U.ptr(x2), U.ptr(w), U.ptr(bias), U.ptr(temb), U.ptr(res), U.ptr(y),
                                       B, H, W, C1, C2, Cout, OH, OW, mode, epi, Cout if temb is not None else 0) == 0

        def timeit():
            run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                run()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / iters
        best = [1e9, 1e9]
        for _ in range(3):
            for k, v in enumerate((va, vb)):
                cur_val[0] = v
                assert lib.dm_set_option(opt.encode(), v) == 0
                best[k] = min(best[k], timeit())
        flops = 2.0 * M * Cout * (taps * Cin + (int(extra.split(":")[1]) + int(extra.split(":")[2]) if extra.startswith("sc:") else 0))
        tot[0] += n * best[0]
        tot[1] += n * best[1]
        print(f"{name:36s} x{n}  A {best[0]:7.3f} ms {flops / best[0] / 1e9:7.1f} TF/s   B {best[1]:7.3f} ms {flops / best[1] / 1e9:7.1f} TF/s   "
              f"B/A {best[1] / best[0]:.3f}", flush=True)
    print(f"per step (these shapes): A {tot[0]:.2f} ms  B {tot[1]:.2f} ms")


if __name__ == "__main__":
    main()
