"""Helpers shared by the GPU parity tests: packing to the engine's layouts and error metrics."""
import ctypes as C

import numpy as np
import torch

from diff_mining_amd import engine as E


def dev():
    return torch.device("cuda", 0)


def f16_randn(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()


def to_nhwc(x):          # [N,C,H,W] -> [N,H,W,C] contiguous
    return x.permute(0, 2, 3, 1).contiguous()


def to_nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def pack_conv3(w):       # [Cout,Cin,3,3] -> [Cout, 9*Cin], k = (tap, cin)
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def pack_geglu(w, b):    # [8C, C] -> quad-interleaved rows (see engine.hip pack_geglu)
    c8 = w.shape[0]
    c4 = c8 // 2
    rho = torch.arange(c8)
    F, q, r = rho // 16, (rho % 16) // 4, rho % 4
    src = torch.where(r < 2, 8 * F + 2 * q + r, c4 + 8 * F + 2 * q + (r - 2))
    return w[src].contiguous(), b[src].contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def op_igemm(X, W, bias=None, X2=None, temb=None, res=None, mode=0, epi=0, OH=None, OW=None, sync=True):
    """X [N,H,W,C1] fp16 cuda; W packed [Cout, taps*Cin] fp16 cuda -> Y [N,OH,OW,Cout(/2)]"""
    lib = E.load_library()
    N, H, Wd, C1 = X.shape
    C2 = X2.shape[3] if X2 is not None else 0
    Cout = W.shape[0]
    OH = H if OH is None else OH
    OW = Wd if OW is None else OW
    cy = Cout // 2 if epi == 1 else Cout
    Y = torch.empty(N, OH, OW, cy, dtype=torch.float16, device=X.device)
    rc = lib.dm_op_igemm(stream(), ptr(X), ptr(X2), ptr(W), ptr(bias), ptr(temb), ptr(res), ptr(Y),
                         N, H, Wd, C1, C2, Cout, OH, OW, mode, epi, temb.stride(0) if temb is not None else 0)
    assert rc == 0, "dm_op_igemm failed"
    if sync:
        torch.cuda.synchronize()
    return Y


def op_attention(Q, K, V, heads, slots=None):
    """Q [B,Tq,C], K/V [Bk,Tk,C] fp16 cuda -> O [B,Tq,C]"""
    lib = E.load_library()
    B, Tq, Cc = Q.shape
    Tk = K.shape[1]
    D = Cc // heads
    O = torch.empty_like(Q)
    rc = lib.dm_op_attention(stream(), ptr(Q), ptr(K), ptr(V), ptr(O), Q.stride(1), K.stride(1), V.stride(1), Cc,
                             Q.stride(0), K.stride(0), V.stride(0), Tq * Cc, ptr(slots), B, heads, Tq, Tk, D,
                             float(D) ** -0.5)
    assert rc == 0, "dm_op_attention failed"
    torch.cuda.synchronize()
    return O


def op_groupnorm(X, gamma, beta, G, eps, silu, X2=None):
    lib = E.load_library()
    N, H, Wd, C1 = X.shape
    Ct = C1 + (X2.shape[3] if X2 is not None else 0)
    Y = torch.empty(N, H, Wd, Ct, dtype=torch.float16, device=X.device)
    rc = lib.dm_op_groupnorm(stream(), ptr(X), ptr(X2), N, H * Wd, Ct, C1, G, eps, ptr(gamma), ptr(beta),
                             1 if silu else 0, ptr(Y))
    assert rc == 0
    torch.cuda.synchronize()
    return Y


def op_layernorm(X, gamma, beta, eps=1e-5):
    lib = E.load_library()
    rows, Cc = X.shape
    Y = torch.empty_like(X)
    rc = lib.dm_op_layernorm(stream(), ptr(X), rows, Cc, ptr(gamma), ptr(beta), eps, ptr(Y))
    assert rc == 0
    torch.cuda.synchronize()
    return Y


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def max_abs(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


def assert_close_fp16(got, ref, what, rel=2e-3, abs_frac=2e-3):
    """fp16-output op vs fp32 reference: rel-L2 and max-abs (as a fraction of max|ref|) bounds."""
    r = rel_l2(got, ref)
    m = max_abs(got, ref) / max(ref.abs().max().item(), 1e-30)
    assert not torch.isnan(got.float()).any().item(), f"{what}: NaN in output"
    assert r < rel and m < abs_frac, f"{what}: rel_l2={r:.3e} (<{rel}) max_abs/max_ref={m:.3e} (<{abs_frac})"
    return r, m


def fold_upconv_torch(w):
    """Upsample2D (nearest 2x) + conv3x3 folded onto the source grid, built with torch: w [Cout,Cin,3,3] ->
    [4 = py*2+px][Cout][(a*2+b)*Cin + ci] fp16; the 3x3 taps that read the same source pixel are summed in fp32, rounded once."""
    sel = {0: ([0], [1, 2]), 1: ([0, 1], [2])}          # parity -> 3x3 taps feeding 2x2 tap a = 0 / 1
    wf = w.float()
    out = []
    for py in (0, 1):
        for px in (0, 1):
            taps = []
            for a in (0, 1):
                for b in (0, 1):
                    taps.append(wf[:, :, sel[py][a], :][:, :, :, sel[px][b]].sum(dim=(2, 3)))     # [Cout, Cin]
            out.append(torch.cat(taps, dim=1))
    return torch.stack(out, 0).half().contiguous()


def op_upconv_folded(X, W4, bias):
    """X [N,H,W,Cin] fp16 cuda, W4 [4][Cout][4*Cin] fp16 cuda -> Y [N,2H,2W,Cout]"""
    lib = E.load_library()
    N, H, Wd, Cin = X.shape
    Cout = W4.shape[1]
    Y = torch.empty(N, 2 * H, 2 * Wd, Cout, dtype=torch.float16, device=X.device)
    rc = lib.dm_op_upconv_folded(stream(), ptr(X), ptr(W4), ptr(bias), ptr(Y), N, H, Wd, Cin, Cout)
    assert rc == 0, "dm_op_upconv_folded failed"
    torch.cuda.synchronize()
    return Y
