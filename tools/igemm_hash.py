#!/usr/bin/env python
"""Prints a sha256 per output of dm_op_igemm / dm_op_igemm_ln on shapes that take the 256 x 320 tile, so that two
builds / runtime switches (e.g. DM_IGEMM_BIG=0 vs 1) can be compared bit for bit from separate processes:

    DM_IGEMM_BIG=0 python tools/igemm_hash.py > a.txt; DM_IGEMM_BIG=1 python tools/igemm_hash.py > b.txt; diff a.txt b.txt
"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_mining_amd import engine as E  # noqa: E402
from tests import gpu_util as U  # noqa: E402


def h(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def main():
    lib = E.load_library()
    d = U.dev()
    g = torch.Generator(device="cuda").manual_seed(7)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=d, dtype=torch.float32) * scale).half()
    # (name, N, H, W, C1, C2, Cout, mode, epi, temb, res)
    cases = [
        ("conv3 640->640 +temb", 160, 32, 32, 640, 0, 640, 1, 0, True, False),
        ("conv3 1280+640->640", 160, 32, 32, 1280, 640, 640, 1, 0, False, False),
        ("conv3 640->640 +res", 160, 32, 32, 640, 0, 640, 1, 0, False, True),
        ("conv3 ragged M", 161, 32, 32, 640, 0, 640, 1, 0, False, True),
        ("conv3 s2 640", 160, 64, 64, 640, 0, 640, 2, 0, False, False),
        ("conv3 up 1280", 160, 16, 16, 1280, 0, 1280, 3, 0, False, False),
        ("dense 1280->1280 +res", 1, 1, 66000, 1280, 0, 1280, 0, 0, False, True),
        ("dense 640->640", 1, 1, 131072 + 100, 640, 0, 640, 0, 0, False, False),
        ("geglu 320->2560", 1, 1, 140000, 320, 0, 2560, 0, 1, False, False),
        ("geglu 1280->10240", 1, 1, 40960, 1280, 0, 10240, 0, 1, False, False),
    ]
    for name, N, H, W, C1, C2, Cout, mode, epi, temb, res in cases:
        taps = 9 if mode else 1
        OH, OW = (H // 2, W // 2) if mode == 2 else ((2 * H, 2 * W) if mode == 3 else (H, W))
        x = rnd(N, H, W, C1)
        x2 = rnd(N, H, W, C2, scale=0.5) if C2 else None
        w = rnd(Cout, taps * (C1 + C2), scale=(taps * (C1 + C2)) ** -0.5)
        b = rnd(Cout, scale=0.1)
        tb = rnd(N, Cout) if temb else None
        rs = rnd(N, OH, OW, Cout) if res else None
        M = N * OH * OW
        tile = lib.dm_op_igemm_tile(M, C1 + C2, Cout, mode)
        y = U.op_igemm(x, w, b, X2=x2, temb=tb, res=rs, mode=mode, epi=epi, OH=OH, OW=OW)
        print(f"{name:28s} tile={tile} {h(y)}", flush=True)
    # LayerNorm-folded linears
    for M, C, Cout, epi in [(131072 + 6, 320, 2560, 1), (66000, 1280, 3840, 0), (655360, 320, 960, 0)]:
        x = rnd(M, C)
        wf = rnd(Cout, C, scale=C ** -0.5)
        ln_s = wf.float().sum(1).contiguous()
        ln_t = (torch.randn(Cout, generator=g, device=d) * 0.1).contiguous()
        stats = torch.empty(M, 2, dtype=torch.float32, device=d)
        assert lib.dm_op_ln_stats(U.stream(), U.ptr(x), M, C, 1e-5, U.ptr(stats)) == 0
        y = torch.empty(M, Cout // 2 if epi else Cout, dtype=torch.float16, device=d)
        assert lib.dm_op_igemm_ln(U.stream(), U.ptr(x), U.ptr(wf), U.ptr(ln_s), U.ptr(ln_t), U.ptr(stats), U.ptr(y), M, C, Cout, epi) == 0
        torch.cuda.synchronize()
        print(f"{'ln M=%d C=%d N=%d epi=%d' % (M, C, Cout, epi):28s} {h(y)}", flush=True)


if __name__ == "__main__":
    main()
