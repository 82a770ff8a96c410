"""Per-op parity: every hand-written gfx950 kernel vs the matching torch.nn.functional op evaluated
in fp32 on the CPU from the same fp16 inputs.  Runs on the GPU box only.

Tolerances: kernels accumulate in fp32 and round the output once to fp16, so the bound is fp16
output rounding: rel-L2 <= 2e-3 and max|err| <= 2e-3 * max|ref| (fp16 eps = 9.8e-4)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests import gpu_util as U  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need a GPU"


@pytest.mark.parametrize("M,K,Cout", [(300, 320, 320), (384, 768, 640), (20, 1280, 1280), (1000, 64, 160)])
def test_igemm_dense_bias_residual(M, K, Cout):
    x = U.f16_randn(1, 1, M, K, seed=1)
    w = U.f16_randn(Cout, K, seed=2, scale=K ** -0.5)
    b = U.f16_randn(Cout, seed=3, scale=0.1)
    r = U.f16_randn(1, 1, M, Cout, seed=4)
    ref = F.linear(x.float().view(M, K), w.float(), b.float())
    d = U.dev()
    y = U.op_igemm(x.to(d), w.to(d), b.to(d))
    U.assert_close_fp16(y.view(M, Cout), ref, "dense+bias")
    y = U.op_igemm(x.to(d), w.to(d), b.to(d), res=r.to(d))
    ref2 = (ref.half().float() + r.float().view(M, Cout))
    U.assert_close_fp16(y.view(M, Cout), ref2, "dense+bias+res")
    y = U.op_igemm(x.to(d), w.to(d))
    U.assert_close_fp16(y.view(M, Cout), F.linear(x.float().view(M, K), w.float()), "dense no bias")


def test_igemm_identity_asymmetric():
    """A = I check with an asymmetric operand: catches row/col swaps in the MFMA C layout."""
    K = Cout = 320
    w = torch.eye(K).half()
    x = (torch.arange(256 * K).view(1, 1, 256, K) % 97).half() / 16
    d = U.dev()
    y = U.op_igemm(x.to(d), w.to(d))
    assert torch.equal(y.cpu().view(256, K), x.view(256, K))


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 9, 7, 64, 160), (1, 16, 16, 320, 320), (3, 8, 8, 128, 640)])
def test_igemm_conv3x3(N, H, W, Cin, Cout):
    x = U.f16_randn(N, Cin, H, W, seed=5)
    w = U.f16_randn(Cout, Cin, 3, 3, seed=6, scale=(9 * Cin) ** -0.5)
    b = U.f16_randn(Cout, seed=7, scale=0.1)
    temb = U.f16_randn(N, Cout + 160, seed=8)
    res = U.f16_randn(N, Cout, H, W, seed=9)
    d = U.dev()
    xg, wg, bg = U.to_nhwc(x).to(d), U.pack_conv3(w).to(d), b.to(d)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    y = U.op_igemm(xg, wg, bg, mode=1)
    U.assert_close_fp16(U.to_nchw(y), ref, "conv3x3")
    # + time-embedding channel add (fp16 add after the fp16 conv output) on a strided table
    tg = temb.to(d)
    y = U.op_igemm(xg, wg, bg, temb=tg[:, 160:], mode=1)   # non-contiguous view: ld = Cout+160
    ref_t = ref.half().float() + temb[:, 160:].float()[:, :, None, None]
    U.assert_close_fp16(U.to_nchw(y), ref_t, "conv3x3+temb")
    y = U.op_igemm(xg, wg, bg, res=U.to_nhwc(res).to(d), mode=1)
    U.assert_close_fp16(U.to_nchw(y), ref.half().float() + res.float(), "conv3x3+res")
    # stride 2, pad 1
    ref_s2 = F.conv2d(x.float(), w.float(), b.float(), stride=2, padding=1)
    y = U.op_igemm(xg, wg, bg, mode=2, OH=ref_s2.shape[2], OW=ref_s2.shape[3])
    U.assert_close_fp16(U.to_nchw(y), ref_s2, "conv3x3 s2")
    # nearest 2x upsample + conv
    up = F.interpolate(x.float(), scale_factor=2.0, mode="nearest")
    ref_up = F.conv2d(up, w.float(), b.float(), padding=1)
    y = U.op_igemm(xg, wg, bg, mode=3, OH=2 * H, OW=2 * W)
    U.assert_close_fp16(U.to_nchw(y), ref_up, "upsample2x+conv3x3")
    # forced output size (odd latent sizes: dift.py:146-147 `upsample_size`)
    oh, ow = 2 * H - 1, 2 * W - 1
    up = F.interpolate(x.float(), size=(oh, ow), mode="nearest")
    ref_up = F.conv2d(up, w.float(), b.float(), padding=1)
    y = U.op_igemm(xg, wg, bg, mode=3, OH=oh, OW=ow)
    U.assert_close_fp16(U.to_nchw(y), ref_up, "upsample(size)+conv3x3")


def test_igemm_concat_sources():
    N, H, W, C1, C2, Cout = 2, 8, 8, 128, 64, 320
    x1, x2 = U.f16_randn(N, C1, H, W, seed=10), U.f16_randn(N, C2, H, W, seed=11)
    w = U.f16_randn(Cout, C1 + C2, 3, 3, seed=12, scale=(9 * (C1 + C2)) ** -0.5)
    w1 = U.f16_randn(Cout, C1 + C2, seed=13, scale=(C1 + C2) ** -0.5)
    d = U.dev()
    cat = torch.cat([x1, x2], 1).float()
    y = U.op_igemm(U.to_nhwc(x1).to(d), U.pack_conv3(w).to(d), X2=U.to_nhwc(x2).to(d), mode=1)
    U.assert_close_fp16(U.to_nchw(y), F.conv2d(cat, w.float(), padding=1), "concat conv3x3")
    y = U.op_igemm(U.to_nhwc(x1).to(d), w1.to(d), X2=U.to_nhwc(x2).to(d), mode=0)
    U.assert_close_fp16(U.to_nchw(y), F.conv2d(cat, w1.float()[:, :, None, None]), "concat 1x1")


def test_igemm_geglu():
    M, Cc = 200, 320
    x = U.f16_randn(1, 1, M, Cc, seed=14)
    w = U.f16_randn(8 * Cc, Cc, seed=15, scale=Cc ** -0.5)
    b = U.f16_randn(8 * Cc, seed=16, scale=0.1)
    proj = F.linear(x.float().view(M, Cc), w.float(), b.float()).half().float()
    a, g = proj.chunk(2, dim=-1)
    ref = a * F.gelu(g).half().float()
    wp, bp = U.pack_geglu(w, b)
    d = U.dev()
    y = U.op_igemm(x.to(d), wp.to(d), bp.to(d), epi=1)
    assert y.shape[-1] == 4 * Cc
    U.assert_close_fp16(y.view(M, 4 * Cc), ref, "geglu", rel=3e-3, abs_frac=3e-3)


def test_geglu_gelu_over_every_fp16_gate():
    """The GELU of the GEGLU epilogue over ALL 63 488 finite fp16 gate values (the gate is rounded to fp16 before the GELU, so this
    is everything the epilogue can ever see): a GEMM whose value column is exactly 1 and whose gate column is exactly x, on the
    persistent 256 x 320 tile and on the 128-row tile.  Against x Phi(x) in float64 rounded to fp16: the Abramowitz-Stegun form
    of the epilogue differs for 257 inputs by one fp16 ulp (tools/gen_gelu_table.py emulates it; a 64-entry cubic table form
    that differs for 4 was built and measured in r04 — same time, not kept: profiles/r04_ab_gelu_lut.txt), and the two tiles
    agree bit for bit."""
    import math
    from scipy.special import erfc
    from diff_mining_amd import engine as E
    lib = E.load_library()
    bits = np.arange(65536, dtype=np.uint16)
    x16 = bits.view(np.float16)
    x16 = x16[np.isfinite(x16)]
    M, K, Cout = len(x16), 256, 640                      # 320 GEGLU outputs
    X = torch.zeros(1, 1, M, K, dtype=torch.float16)
    X[0, 0, :, 0] = torch.from_numpy(x16.copy())
    X[0, 0, :, 1] = 1.0
    w = torch.zeros(Cout, K, dtype=torch.float16)
    w[: Cout // 2, 1] = 1.0                              # value half: h = 1
    w[Cout // 2:, 0] = 1.0                               # gate half: g = x
    b = torch.zeros(Cout, dtype=torch.float16)
    wp, bp = U.pack_geglu(w, b)
    d = U.dev()
    out = {}
    try:
        for big in (0, 1):
            assert lib.dm_set_option(b"igemm_big", big) == 0
            out[big] = U.op_igemm(X.to(d), wp.to(d), bp.to(d), epi=1).view(M, Cout // 2).cpu()
    finally:
        lib.dm_set_option(b"igemm_big", -1)
    assert torch.equal(out[0], out[1]), "the two tile kernels evaluate the GELU differently"
    got = out[1].numpy()
    assert (got == got[:, :1]).all()                     # every output channel sees the same gate
    x64 = x16.astype(np.float64)
    exact = x64 * (1.0 - 0.5 * erfc(x64 / math.sqrt(2.0)))
    ref16 = exact.astype(np.float16)
    diff = got[:, 0] != ref16
    n, big_n = int(diff.sum()), int((diff & (np.abs(exact) > 1e-4)).sum())
    worst = float(np.abs(got[:, 0].astype(np.float64) - exact).max())
    print(f"GELU over {M} fp16 gates: results that differ from the correctly rounded x Phi(x): {n} ({big_n} with |gelu| > 1e-4); "
          f"max |fp16 result - float64| {worst:.2e}")
    assert n <= 300, n


@pytest.mark.parametrize("D,Tq,Tk", [(40, 256, 256), (40, 200, 200), (80, 336, 336), (160, 64, 64), (40, 4096, 4096), (80, 1024, 1024), (80, 256, 256)])
def test_attention_self(D, Tq, Tk):
    heads, B = 8, 2
    Cc = heads * D
    qkv = U.f16_randn(B, Tq, 3 * Cc, seed=17)
    q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]

    def split(t):
        return t.float().view(B, -1, heads, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(B, Tq, Cc)
    g = qkv.to(U.dev())
    o = U.op_attention(g[..., :Cc], g[..., Cc:2 * Cc], g[..., 2 * Cc:], heads)   # strided views of fused QKV
    U.assert_close_fp16(o, ref, f"self-attn D={D} T={Tq}", rel=3e-3, abs_frac=4e-3)


@pytest.mark.parametrize("D", [40, 80, 160])
def test_attention_cross_with_prompt_slots(D):
    heads, B, Tq, Tk, P = 8, 5, 150, 77, 3
    Cc = heads * D
    q = U.f16_randn(B, Tq, Cc, seed=18)
    kv = U.f16_randn(P, Tk, 2 * Cc, seed=19)
    slots = torch.tensor([2, 0, 1, 1, 2], dtype=torch.int32)
    k, v = kv[..., :Cc][slots.long()], kv[..., Cc:][slots.long()]

    def split(t, T):
        return t.float().view(B, T, heads, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(split(q, Tq), split(k, Tk), split(v, Tk)).transpose(1, 2).reshape(B, Tq, Cc)
    d = U.dev()
    kvg = kv.to(d)
    o = U.op_attention(q.to(d), kvg[..., :Cc], kvg[..., Cc:], heads, slots=slots.to(d))
    U.assert_close_fp16(o, ref, f"cross-attn D={D}", rel=3e-3, abs_frac=4e-3)


@pytest.mark.parametrize("D,Tq,B", [(40, 4096, 4), (40, 1000, 5), (80, 1024, 5), (80, 300, 3)])
def test_attention_cross_resident_kv_kernel(D, Tq, B):
    """The 77-key cross-attention kernel that keeps K / V of a (prompt, head) resident in LDS over all query blocks of a
    sample (head_dim 40 / 80, Tq >= 256): ragged query counts, prompt slots, a dominating key; against fp32 SDPA and the
    generic kernel (`attn_cross` = 0)."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    heads, Tk, P = 8, 77, 3
    Cc = heads * D
    q = U.f16_randn(B, Tq, Cc, seed=28)
    kv = U.f16_randn(P, Tk, 2 * Cc, seed=29)
    kv[1, 70, :Cc] = q[1, 9] * 3.0                    # one late key dominates some rows of prompt 1
    slots = torch.tensor([2, 1, 0, 1, 2][:B], dtype=torch.int32)
    k, v = kv[..., :Cc][slots.long()], kv[..., Cc:][slots.long()]

    def split(t, T):
        return t.float().view(B, T, heads, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(split(q, Tq), split(k, Tk), split(v, Tk)).transpose(1, 2).reshape(B, Tq, Cc)
    d = U.dev()
    qd, kvg, sl = q.to(d), kv.to(d), slots.to(d)
    try:
        o = U.op_attention(qd, kvg[..., :Cc], kvg[..., Cc:], heads, slots=sl)
        assert lib.dm_set_option(b"attn_cross", 0) == 0
        o_gen = U.op_attention(qd, kvg[..., :Cc], kvg[..., Cc:], heads, slots=sl)
    finally:
        lib.dm_set_option(b"attn_cross", 1)
    U.assert_close_fp16(o, ref, f"cross-attn resident D={D} Tq={Tq}", rel=3e-3, abs_frac=4e-3)
    U.assert_close_fp16(o, o_gen.float().cpu(), f"cross-attn resident vs generic D={D}", rel=3e-3, abs_frac=4e-3)


def test_attention_large_logits_online_softmax():
    """Force the running-max rescale path: one key per row dominates at a late tile."""
    heads, B, T, D = 8, 1, 320, 40
    Cc = heads * D
    q = U.f16_randn(B, T, Cc, seed=20)
    k = U.f16_randn(B, T, Cc, seed=21)
    v = U.f16_randn(B, T, Cc, seed=22)
    k[:, 300] = q[:, 5] * 4.0          # spikes q.k at key 300 (tile 4) for query 5 and correlated rows

    def split(t):
        return t.float().view(B, T, heads, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(B, T, Cc)
    d = U.dev()
    o = U.op_attention(q.to(d), k.to(d), v.to(d), heads)
    U.assert_close_fp16(o, ref, "attn spike", rel=3e-3, abs_frac=4e-3)


@pytest.mark.parametrize("D,Tq,Tk,spike", [(80, 300, 384, None), (80, 130, 512, 400), (40, 300, 384, 300), (80, 4096, 4096, 3000)])
def test_attention_pipelined_kernels_ragged_queries_and_rescale(D, Tq, Tk, spike):
    """The software-pipelined kernels (head_dim 40 and 80; Tk % 128 == 0, >= 256) with a query count that is not a
    multiple of the 128-query block, K/V of another length than Q, and a late dominating key that forces the lazy
    running-max rescale; checked against fp32 SDPA and against the un-pipelined kernel (`attn_pipe` = 0)."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    heads, B = 8, 2
    Cc = heads * D
    q = U.f16_randn(B, Tq, Cc, seed=23)
    k = U.f16_randn(B, Tk, Cc, seed=24)
    v = U.f16_randn(B, Tk, Cc, seed=25)
    if spike is not None:
        k[:, spike] = q[:, 7] * 4.0

    def split(t, T):
        return t.float().view(B, T, heads, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(split(q, Tq), split(k, Tk), split(v, Tk)).transpose(1, 2).reshape(B, Tq, Cc)
    d = U.dev()
    qd, kd, vd = q.to(d), k.to(d), v.to(d)
    try:
        o = U.op_attention(qd, kd, vd, heads)
        assert lib.dm_set_option(b"attn_pipe", 0) == 0
        o_plain = U.op_attention(qd, kd, vd, heads)
    finally:
        lib.dm_set_option(b"attn_pipe", 1)
    U.assert_close_fp16(o, ref, f"pipelined attn D={D} Tq={Tq} Tk={Tk}", rel=3e-3, abs_frac=4e-3)
    U.assert_close_fp16(o, o_plain.float().cpu(), f"pipelined vs plain D={D}", rel=3e-3, abs_frac=4e-3)


@pytest.mark.parametrize("Tq,Tk,spike", [(4096, 4096, 3000), (300, 384, 300), (256, 256, None), (1000, 256, 100), (512, 1024, -1), (130, 512, 400)])
def test_attention_scores_on_32x32_mfma(Tq, Tk, spike):
    """attention_qk32.hip (r06, head_dim 40): S^T = K Q'^T on v_mfma_f32_32x32x16_f16 (k = 48 instead of 64), P moved to the PV operand
    layout by v_permlane16_swap, V rows / K chunks permuted in LDS.  Ragged query counts, K/V longer and shorter than Q, the minimum of
    four tiles, a late dominating key (lazy rescale: the factor of the OTHER 16-query block travels 16 lanes), all-negative logits;
    against fp32 SDPA and the kernel it replaces; deterministic; batch-independent."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    heads, B, D = 8, 2, 40
    Cc = heads * D
    q = U.f16_randn(B, Tq, Cc, seed=31)
    k = U.f16_randn(B, Tk, Cc, seed=32)
    v = U.f16_randn(B, Tk, Cc, seed=33)
    if spike is not None and spike >= 0:
        k[:, spike] = q[:, 7] * 4.0                              # query 7 (and whoever correlates) meets a dominating key late
        k[:, spike - 70, :40] = q[:, 21, :40] * 3.0              # ... and query 21 of head 0 (the other 16-query block) one tile earlier
    if spike == -1:
        k, q = -k.abs(), q.abs()                                 # every logit negative: the first tile must set the running max

    def split(t, T):
        return t.float().view(B, T, heads, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(split(q, Tq), split(k, Tk), split(v, Tk)).transpose(1, 2).reshape(B, Tq, Cc)
    d = U.dev()
    qd, kd, vd = q.to(d), k.to(d), v.to(d)
    try:
        assert lib.dm_set_option(b"attn_pipe", 5) == 0
        o = U.op_attention(qd, kd, vd, heads)
        o2 = U.op_attention(qd, kd, vd, heads)
        o1 = U.op_attention(qd[1:2].contiguous(), kd[1:2].contiguous(), vd[1:2].contiguous(), heads)
        assert lib.dm_set_option(b"attn_pipe", 9) == 0
        o_pipe = U.op_attention(qd, kd, vd, heads)
    finally:
        lib.dm_set_option(b"attn_pipe", 1)
    e32, e16 = U.rel_l2(o, ref), U.rel_l2(o_pipe, ref)
    print(f"qk32 attention Tq={Tq} Tk={Tk} spike={spike}: rel-L2 vs fp32 SDPA {e32:.2e} (attn_pipe_kernel {e16:.2e})")
    U.assert_close_fp16(o, ref, f"qk32 attn Tq={Tq} Tk={Tk}", rel=3e-3, abs_frac=4e-3)
    assert e32 <= 1.1 * e16 + 2e-5                               # no further from fp32 than the kernel it replaces
    assert torch.equal(o, o2) and torch.equal(o1[0], o[1])


@pytest.mark.parametrize("Tq,Tk,spike,B", [(256, 256, None, 3), (64, 64, None, 3), (300, 128, 100, 3), (256, 64, None, 3), (1024, 1024, 900, 3), (40, 192, -1, 3),
                                            (256, 256, 200, 40), (256, 256, None, 32), (64, 64, None, 64)])
def test_attention_head_dim_160_eight_wave_kernel(Tq, Tk, spike, B):
    """attention_d160.hip (r06): head_dim 160 with eight waves sharing a (sample, head)'s K / V, K chunks / V blocks permuted in LDS.  One
    tile (8x8 level), the 16x16 level, ragged and short query counts (waves without a query), a late dominating key, all-negative logits;
    against fp32 SDPA, and BIT-EQUAL to the generic kernel (the same arithmetic in the same order: only the schedule and the LDS image differ).
    B = 40 / 32 / 64: the persistent blocks walk several units each, in the order that keeps a sample's heads on one XCD (B % 8 == 0 and at
    least one unit per block); B = 3: the plain order."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    heads, D = 8, 160
    Cc = heads * D
    q = U.f16_randn(B, Tq, Cc, seed=41)
    k = U.f16_randn(B, Tk, Cc, seed=42)
    v = U.f16_randn(B, Tk, Cc, seed=43)
    if spike is not None and spike >= 0:
        k[:, spike] = q[:, 7] * 2.0
    if spike == -1:
        k, q = -k.abs(), q.abs()

    def split(t, T):
        return t.float().view(B, T, heads, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(split(q, Tq), split(k, Tk), split(v, Tk)).transpose(1, 2).reshape(B, Tq, Cc)
    d = U.dev()
    qd, kd, vd = q.to(d), k.to(d), v.to(d)
    try:
        assert lib.dm_set_option(b"attn_pipe", 1) == 0
        o = U.op_attention(qd, kd, vd, heads)
        o1 = U.op_attention(qd[2:3].contiguous(), kd[2:3].contiguous(), vd[2:3].contiguous(), heads)
        assert lib.dm_set_option(b"attn_pipe", 9) == 0
        o_gen = U.op_attention(qd, kd, vd, heads)
    finally:
        lib.dm_set_option(b"attn_pipe", 1)
    print(f"head_dim 160 Tq={Tq} Tk={Tk} spike={spike}: rel-L2 vs fp32 SDPA {U.rel_l2(o, ref):.2e}")
    U.assert_close_fp16(o, ref, f"d160 attn Tq={Tq} Tk={Tk}", rel=3e-3, abs_frac=4e-3)
    assert torch.equal(o, o_gen), f"not bit-equal to the generic kernel: max |diff| {(o.float() - o_gen.float()).abs().max().item():.3e}"
    assert torch.equal(o1[0], o[2])


@pytest.mark.parametrize("variant", [10, 12])
@pytest.mark.parametrize("Tq,Tk,spike", [(4096, 4096, 3000), (300, 384, 300), (256, 320, None), (1000, 256, 100), (512, 1024, -1)])
def test_attention_antiphase_kernel(variant, Tq, Tk, spike):
    """attention_pp.hip (head_dim 40; two wave sets per SIMD half an iteration apart): ragged query counts, key counts that are
    odd multiples of the 64-key tile, the minimum of four tiles, a late dominating key (lazy rescale), and all-negative logits
    (spike = -1: the first tile must set the running max whatever its sign); against fp32 SDPA and the generic kernel."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    heads, B, D = 8, 2, 40
    Cc = heads * D
    q = U.f16_randn(B, Tq, Cc, seed=33)
    k = U.f16_randn(B, Tk, Cc, seed=34)
    v = U.f16_randn(B, Tk, Cc, seed=35)
    if spike == -1:
        q = (q.float().abs() * 3.0).half()
        k = (-k.float().abs() * 3.0).half()               # every logit strongly negative: sc * q.k ~ -500
    elif spike is not None:
        k[:, spike] = q[:, 7] * 4.0

    def split(t, T):
        return t.float().view(B, T, heads, D).transpose(1, 2)
    ref = F.scaled_dot_product_attention(split(q, Tq), split(k, Tk), split(v, Tk)).transpose(1, 2).reshape(B, Tq, Cc)
    d = U.dev()
    qd, kd, vd = q.to(d), k.to(d), v.to(d)
    try:
        assert lib.dm_set_option(b"attn_pipe", variant) == 0
        o = U.op_attention(qd, kd, vd, heads)
        o2 = U.op_attention(qd, kd, vd, heads)
        assert lib.dm_set_option(b"attn_pipe", 0) == 0
        o_plain = U.op_attention(qd, kd, vd, heads)
    finally:
        lib.dm_set_option(b"attn_pipe", 1)
    assert torch.equal(o, o2), "not deterministic"
    U.assert_close_fp16(o, ref, f"anti-phase attn v{variant} Tq={Tq} Tk={Tk}", rel=3e-3, abs_frac=4e-3)
    U.assert_close_fp16(o, o_plain.float().cpu(), f"anti-phase vs plain v{variant}", rel=3e-3, abs_frac=4e-3)


@pytest.mark.parametrize("C1,C2,silu,eps", [(320, 0, True, 1e-5), (640, 0, False, 1e-6), (1280, 640, True, 1e-5),
                                            (640, 320, True, 1e-5), (1280, 1280, True, 1e-5)])
def test_groupnorm(C1, C2, silu, eps):
    N, H, W = 3, 9, 7
    x1 = U.f16_randn(N, C1, H, W, seed=23) * 2 + 0.5
    x2 = (U.f16_randn(N, C2, H, W, seed=24) * 0.5 - 1.0) if C2 else None
    Ct = C1 + C2
    g = torch.randn(Ct, generator=torch.Generator().manual_seed(25)) * 0.1 + 1
    b = torch.randn(Ct, generator=torch.Generator().manual_seed(26)) * 0.1
    cat = torch.cat([x1, x2], 1).float() if C2 else x1.float()
    ref = F.group_norm(cat, 32, g, b, eps)
    if silu:
        ref = F.silu(ref)
    d = U.dev()
    y = U.op_groupnorm(U.to_nhwc(x1).to(d), g.to(d), b.to(d), 32, eps, silu,
                       X2=U.to_nhwc(x2).to(d) if C2 else None)
    U.assert_close_fp16(U.to_nchw(y), ref, f"groupnorm C={Ct}")


@pytest.mark.parametrize("Cc", [320, 640, 1280])
def test_layernorm(Cc):
    rows = 777
    x = U.f16_randn(rows, Cc, seed=27) * 3 + 1
    g = torch.randn(Cc, generator=torch.Generator().manual_seed(28)) * 0.1 + 1
    b = torch.randn(Cc, generator=torch.Generator().manual_seed(29)) * 0.1
    ref = F.layer_norm(x.float(), (Cc,), g, b, 1e-5)
    d = U.dev()
    y = U.op_layernorm(x.to(d), g.to(d), b.to(d))
    U.assert_close_fp16(y, ref, f"layernorm C={Cc}")


@pytest.mark.parametrize("M,Cout,epi", [(131072, 320, 0), (131072 + 77, 960, 0), (140000, 2560, 1), (655360, 320, 0)])
def test_igemm_k320_many_rows(M, Cout, epi):
    """The K = 320 dense layers of the 64x64 transformer blocks at production row counts (>= 128 Ki rows, ragged
    last row tile): bias, residual, GEGLU."""
    K = 320
    x = U.f16_randn(1, 1, M, K, seed=1)
    w = U.f16_randn(Cout, K, seed=2, scale=K ** -0.5)
    b = U.f16_randn(Cout, seed=3, scale=0.1)
    d = U.dev()
    if epi == 1:
        wp, bp = U.pack_geglu(w, b)
        y = U.op_igemm(x.to(d), wp.to(d), bp.to(d), epi=1).view(M, Cout // 2)
        rows = torch.cat([torch.arange(0, 4096), torch.arange(M - 4096, M)])
        hg = F.linear(x.view(M, K)[rows].float(), w.float(), b.float()).half().float()
        ref = (hg[:, :Cout // 2] * F.gelu(hg[:, Cout // 2:]).half().float())
        U.assert_close_fp16(y[rows.to(d)], ref, "k320 geglu", rel=3e-3, abs_frac=4e-3)
    else:
        r = U.f16_randn(1, 1, M, Cout, seed=4)
        y = U.op_igemm(x.to(d), w.to(d), b.to(d), res=r.to(d)).view(M, Cout)
        rows = torch.cat([torch.arange(0, 4096), torch.arange(M - 4096, M)])
        ref = F.linear(x.view(M, K)[rows].float(), w.float(), b.float()).half().float() + r.view(M, Cout)[rows].float()
        U.assert_close_fp16(y[rows.to(d)], ref, "k320 dense+bias+res")
        y2 = U.op_igemm(x.to(d), w.to(d)).view(M, Cout)
        U.assert_close_fp16(y2[rows.to(d)], F.linear(x.view(M, K)[rows].float(), w.float()), "k320 dense no bias")


@pytest.mark.parametrize("N,H,W,Cin,C2,Cout,mode,ks", [(40, 8, 8, 1280, 0, 1280, 1, 4), (16, 8, 8, 1280, 1280, 1280, 1, 8),
                                                       (3, 9, 7, 640, 0, 320, 1, 3), (1, 1, 500, 2560, 0, 640, 0, 4)])
def test_igemm_splitk_matches_unsplit(N, H, W, Cin, C2, Cout, mode, ks):
    """Split-K (fp32 partial tiles + reduction/epilogue kernel) vs the fused single-pass kernel: same rounding
    points (bias -> fp16, + time embedding -> fp16, + residual -> fp16), so results agree to fp32 summation order."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    taps = 9 if mode else 1
    x = U.f16_randn(N, H, W, Cin, seed=1).to(d)
    x2 = U.f16_randn(N, H, W, C2, seed=2).to(d) if C2 else None
    w = U.f16_randn(Cout, taps * (Cin + C2), seed=3, scale=(taps * (Cin + C2)) ** -0.5).to(d)
    b = U.f16_randn(Cout, seed=4, scale=0.1).to(d)
    temb = U.f16_randn(N, Cout, seed=5).to(d) if mode else None
    res = U.f16_randn(N, H, W, Cout, seed=6).to(d)
    ref = U.op_igemm(x, w, b, X2=x2, temb=temb, res=res, mode=mode)
    y = torch.empty_like(ref)
    ws = torch.empty(ks * N * H * W * Cout, dtype=torch.float32, device=d)
    rc = lib.dm_op_igemm_splitk(U.stream(), U.ptr(x), U.ptr(x2), U.ptr(w), U.ptr(b), U.ptr(temb), U.ptr(res), U.ptr(y),
                                N, H, W, Cin, C2, Cout, H, W, mode, temb.stride(0) if temb is not None else 0, ks, U.ptr(ws))
    assert rc == 0
    torch.cuda.synchronize()
    diff = (y.float() - ref.float()).abs()
    # identical except where the fp32 summation order flips an fp16 rounding (1 ulp, rare)
    assert (diff > 0).float().mean().item() < 0.02 and diff.max().item() <= 2e-3 * ref.float().abs().max().item()


@pytest.mark.parametrize("N,H,W,Cin,C2,Cout", [(160, 8, 8, 1280, 0, 1280), (160, 8, 8, 1280, 1280, 1280), (157, 8, 8, 640, 0, 1280), (100, 9, 7, 640, 320, 640)])
def test_persistent_splitk_is_bit_identical(N, H, W, Cin, C2, Cout):
    """Split-K on the persistent 256 x 320 tile (three parts of three taps: (tile, part) units, fp32 partials, the common
    reduction kernel) against the 128-row split-K kernel with the same three parts (`igemm_splitk` = 2 keeps every split
    launch on the 128-row tile): same k order inside a part, same reduction -> the same bits, ragged last row tile and
    channel concat included; and both agree with the unsplit kernel to fp32 summation order."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    x = U.f16_randn(N, H, W, Cin, seed=1).to(d)
    x2 = U.f16_randn(N, H, W, C2, seed=2).to(d) if C2 else None
    w = U.f16_randn(Cout, 9 * (Cin + C2), seed=3, scale=(9 * (Cin + C2)) ** -0.5).to(d)
    b = U.f16_randn(Cout, seed=4, scale=0.1).to(d)
    temb = U.f16_randn(N, Cout, seed=5).to(d)
    res = U.f16_randn(N, H, W, Cout, seed=6).to(d)
    ref = U.op_igemm(x, w, b, X2=x2, temb=temb, res=res, mode=1)
    ws = torch.empty(3 * N * H * W * Cout, dtype=torch.float32, device=d)

    def run():
        y = torch.full_like(ref, float("nan"))
        ws.fill_(float("nan"))
        assert lib.dm_op_igemm_splitk(U.stream(), U.ptr(x), U.ptr(x2), U.ptr(w), U.ptr(b), U.ptr(temb), U.ptr(res), U.ptr(y),
                                      N, H, W, Cin, C2, Cout, H, W, 1, temb.stride(0), 3, U.ptr(ws)) == 0
        torch.cuda.synchronize()
        return y
    try:
        y_pers = run()
        y_again = run()
        assert lib.dm_set_option(b"igemm_splitk", 2) == 0
        y_small = run()
    finally:
        lib.dm_set_option(b"igemm_splitk", 1)
    assert not torch.isnan(y_pers.float()).any()
    assert torch.equal(y_pers, y_small) and torch.equal(y_pers, y_again)
    diff = (y_pers.float() - ref.float()).abs()
    assert (diff > 0).float().mean().item() < 0.02 and diff.max().item() <= 2e-3 * ref.float().abs().max().item()


@pytest.mark.parametrize("M,C,Cout,epi", [(300, 320, 960, 0), (4096, 640, 640, 0), (131072 + 5, 320, 2560, 1), (8192, 1280, 10240, 1),
                                          (655360, 320, 320, 0)])
def test_layernorm_folded_into_linear(M, C, Cout, epi):
    """LN -> Linear (-> GEGLU) with the LayerNorm folded into the GEMM (weights scaled by gamma, per-row mean / rstd
    correction in the epilogue) vs F.linear(F.layer_norm(x)) in fp32; covers the 128x320 / 128x160 and 256x320 tiles."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    x = (U.f16_randn(M, C, seed=1).float() * 1.5 + 0.7 * U.f16_randn(M, 1, seed=2).float()).half()   # rows with non-zero mean
    w = U.f16_randn(Cout, C, seed=3, scale=C ** -0.5)
    b = U.f16_randn(Cout, seed=4, scale=0.1)
    gamma = (1 + 0.1 * U.f16_randn(C, seed=5).float()).half()
    beta = (0.05 * U.f16_randn(C, seed=6).float()).half()
    if epi == 1:
        wq, bq = U.pack_geglu(w, b)
    else:
        wq, bq = w, b
    wf = (wq.float() * gamma.float()[None]).half()
    ln_s = wf.float().sum(1)
    ln_t = (wq.float() @ beta.float()) + bq.float()
    xg = x.to(d)
    stats = torch.empty(M, 2, dtype=torch.float32, device=d)
    assert lib.dm_op_ln_stats(U.stream(), U.ptr(xg), M, C, 1e-5, U.ptr(stats)) == 0
    y = torch.empty(M, Cout // 2 if epi else Cout, dtype=torch.float16, device=d)
    wfd, sd_, td_ = wf.to(d), ln_s.to(d), ln_t.to(d)
    assert lib.dm_op_igemm_ln(U.stream(), U.ptr(xg), U.ptr(wfd), U.ptr(sd_), U.ptr(td_), U.ptr(stats), U.ptr(y), M, C, Cout, epi) == 0
    torch.cuda.synchronize()
    rows = torch.arange(M) if M <= 8192 else torch.cat([torch.arange(0, 2048), torch.arange(M - 2048, M)])
    xr = x[rows].float()
    st = stats[rows.to(d)].cpu()
    assert torch.allclose(st[:, 0], xr.mean(1), atol=1e-5) and torch.allclose(st[:, 1], (xr.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4)
    h = F.linear(F.layer_norm(xr, (C,), gamma.float(), beta.float(), 1e-5), w.float(), b.float())
    if epi == 1:
        hh = h.half().float()
        ref = hh[:, :Cout // 2] * F.gelu(hh[:, Cout // 2:]).half().float()
    else:
        ref = h
    U.assert_close_fp16(y[rows.to(d)], ref, f"LN-folded linear epi={epi}", rel=3e-3, abs_frac=4e-3)


@pytest.mark.parametrize("C1,C2,Cout", [(640, 0, 640), (1280, 640, 640), (1280, 0, 1280)])
def test_igemm_conv3x3_big_tile_vs_conv2d(C1, C2, Cout):
    """The 256 px x 320 ch tile on its 3x3-convolution shapes (>= 1024 big tiles, Cin >= 640: the 32x32 layers of
    up_blocks[1..2] at the bench batch, 25 % of a step's GPU time), checked DIRECTLY against F.conv2d in fp32 on whole
    sampled images (first / middle / last of the batch, so row tiles that straddle images are covered), with the
    channel concat, the time-embedding add and the residual epilogue."""
    from diff_mining_amd import engine as E
    N, H, W = 160, 32, 32
    lib = E.load_library()
    assert lib.dm_op_igemm_tile(N * H * W, C1 + C2, Cout, 1) == 1, "shape is not routed to the 256x320 tile"
    d = U.dev()
    Cin = C1 + C2
    g = torch.Generator(device="cuda").manual_seed(31)
    x1 = torch.randn(N, H, W, C1, generator=g, device=d, dtype=torch.float32).half()
    x2 = (torch.randn(N, H, W, C2, generator=g, device=d, dtype=torch.float32) * 0.5).half() if C2 else None
    w = U.f16_randn(Cout, Cin, 3, 3, seed=32, scale=(9 * Cin) ** -0.5)
    b = U.f16_randn(Cout, seed=33, scale=0.1)
    temb = U.f16_randn(N, Cout, seed=34)
    res = torch.randn(N, H, W, Cout, generator=g, device=d, dtype=torch.float32).half()
    wg, bg = U.pack_conv3(w).to(d), b.to(d)
    y_plain = U.op_igemm(x1, wg, bg, X2=x2, mode=1)
    y_temb = U.op_igemm(x1, wg, bg, X2=x2, temb=temb.to(d), mode=1)
    y_res = U.op_igemm(x1, wg, bg, X2=x2, res=res, mode=1)
    for n in (0, 77, N - 1):
        xin = x1[n:n + 1] if x2 is None else torch.cat([x1[n:n + 1], x2[n:n + 1]], 3)
        ref = F.conv2d(U.to_nchw(xin.cpu().float()), w.float(), b.float(), padding=1)
        U.assert_close_fp16(U.to_nchw(y_plain[n:n + 1]), ref, f"big conv3x3 n={n}")
        U.assert_close_fp16(U.to_nchw(y_temb[n:n + 1]), ref.half().float() + temb[n].float()[None, :, None, None],
                            f"big conv3x3+temb n={n}")
        U.assert_close_fp16(U.to_nchw(y_res[n:n + 1]), ref.half().float() + U.to_nchw(res[n:n + 1].cpu().float()),
                            f"big conv3x3+res n={n}")
    # the 128-row tile on the same operands agrees to fp32 summation order (different k-step grouping)
    small = U.op_igemm(x1[:8], wg, bg, X2=x2[:8] if C2 else None, mode=1)
    assert lib.dm_op_igemm_tile(8 * H * W, Cin, Cout, 1) == 0
    diff = (small.float() - y_plain[:8].float()).abs()
    assert diff.max().item() <= 2e-3 * y_plain[:8].float().abs().max().item()


@pytest.mark.parametrize("case", ["conv_temb", "conv_cat", "conv_res_ragged", "conv_s2", "conv_up", "dense_res", "geglu", "ln_geglu", "ln_plain"])
def test_persistent_tile_is_bit_identical(case):
    """The persistent 256 x 320 tile (continuous k stream across tiles, direct fragment stores through
    v_permlane16/32_swap, dynamic tile hand-out) against the one-tile-per-block 128-row kernel with its LDS-staged
    epilogue: same arithmetic, same rounding points -> equal bit for bit on every epilogue variant, ragged last row tile
    included (which is what makes a sample's result independent of the batch size that selects the tile).
    `dm_set_option` switches kernels inside one process (igemm_big: 0 = never, 1 = every eligible shape, -1 = per shape)."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    g = torch.Generator(device="cuda").manual_seed(11)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=d, dtype=torch.float32) * scale).half()
    ln = case.startswith("ln_")
    N, H, W, C1, C2, Cout, mode, epi, temb, res = {
        "conv_temb": (160, 32, 32, 640, 0, 640, 1, 0, True, False),
        "conv_cat": (160, 32, 32, 1280, 640, 640, 1, 0, True, False),
        "conv_res_ragged": (161, 32, 32, 640, 0, 640, 1, 0, False, True),       # M = 164 864 + ... : last tile partly beyond M
        "conv_s2": (160, 64, 64, 640, 0, 640, 2, 0, False, False),
        "conv_up": (160, 16, 16, 1280, 0, 1280, 3, 0, False, False),
        "dense_res": (1, 1, 66000 + 37, 1280, 0, 1280, 0, 0, False, True),
        "geglu": (1, 1, 140000, 320, 0, 2560, 0, 1, False, False),
        "ln_geglu": (1, 1, 131072 + 6, 320, 0, 2560, 0, 1, False, False),
        "ln_plain": (1, 1, 66000, 1280, 0, 3840, 0, 0, False, False),
    }[case]
    taps = 9 if mode else 1
    OH, OW = (H // 2, W // 2) if mode == 2 else ((2 * H, 2 * W) if mode == 3 else (H, W))
    M = N * OH * OW
    assert lib.dm_op_igemm_tile(M, C1 + C2, Cout, mode) == 1
    x = rnd(N, H, W, C1)
    x2 = rnd(N, H, W, C2, scale=0.5) if C2 else None
    w = rnd(Cout, taps * (C1 + C2), scale=(taps * (C1 + C2)) ** -0.5)
    b = rnd(Cout, scale=0.1)
    tb = rnd(N, Cout) if temb else None
    rs = rnd(N, OH, OW, Cout) if res else None
    if ln:
        ln_s = w.float().sum(1).contiguous()
        ln_t = (torch.randn(Cout, generator=g, device=d) * 0.1).contiguous()
        stats = torch.empty(M, 2, dtype=torch.float32, device=d)
        assert lib.dm_op_ln_stats(U.stream(), U.ptr(x), M, C1, 1e-5, U.ptr(stats)) == 0

    def run():
        if ln:
            y = torch.full((M, Cout // 2 if epi else Cout), float("nan"), dtype=torch.float16, device=d)
            assert lib.dm_op_igemm_ln(U.stream(), U.ptr(x), U.ptr(w), U.ptr(ln_s), U.ptr(ln_t), U.ptr(stats), U.ptr(y), M, C1, Cout, epi) == 0
            torch.cuda.synchronize()
            return y
        return U.op_igemm(x, w, b, X2=x2, temb=tb, res=rs, mode=mode, epi=epi, OH=OH, OW=OW)
    try:
        assert lib.dm_set_option(b"igemm_big", 0) == 0
        assert lib.dm_op_igemm_tile(M, C1 + C2, Cout, mode) == 0
        y0 = run()
        assert lib.dm_set_option(b"igemm_big", 1) == 0
        assert lib.dm_op_igemm_tile(M, C1 + C2, Cout, mode) == 1
        y1 = run()
        y2 = run()                      # a second launch: the self-resetting tile counters left a clean state
    finally:
        lib.dm_set_option(b"igemm_big", -1)
    assert not torch.isnan(y1.float()).any()
    assert torch.equal(y0, y1) and torch.equal(y1, y2)
    assert lib.dm_set_option(b"no_such_option", 1) != 0


def test_persistent_tile_fuzz_against_128_row_tile():
    """Randomised shapes (every mode, concat splits, ragged M, odd spatial sizes, K from 4 k steps up, optional bias /
    time embedding / residual) through both tile kernels: bit-identical wherever the persistent kernel accepts the shape
    (igemm_big = 1 falls back to the 128-row tile where it does not, e.g. a time embedding with HW % 256 != 0)."""
    import random
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    rng = random.Random(1234)
    g = torch.Generator(device="cuda").manual_seed(99)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=d, dtype=torch.float32) * scale).half()
    n_pers = 0
    try:
        for it in range(36):
            mode = rng.choice([0, 0, 1, 1, 1, 2, 3])
            Cout = rng.choice([320, 640, 960, 1280])
            c_tot = rng.choice([256, 320, 384, 640, 960])
            C2 = rng.choice([0, 0, 64, 128, 320]) if c_tot > 320 else 0
            C1 = c_tot - C2
            if mode == 0:
                N, H, W = 1, 1, rng.choice([256, 300, 1000, 1025, 2048, 4097])
            else:
                N = rng.choice([1, 2, 3, 5])
                H, W = rng.choice([(16, 16), (16, 32), (9, 7), (12, 10), (32, 32), (24, 16)])
            OH, OW = (H, W) if mode in (0, 1) else (((H + 1) // 2, (W + 1) // 2) if mode == 2 else
                                                    rng.choice([(2 * H, 2 * W), (2 * H - 1, 2 * W - 1)]))
            epi = 1 if (mode == 0 and rng.random() < 0.25) else 0
            extra = rng.choice(["", "", "temb", "res"]) if not epi else ""
            if extra == "temb" and mode == 0:
                extra = ""
            taps = 9 if mode else 1
            x = rnd(N, H, W, C1)
            x2 = rnd(N, H, W, C2, scale=0.5) if C2 else None
            w = rnd(Cout, taps * c_tot, scale=(taps * c_tot) ** -0.5)
            b = rnd(Cout, scale=0.1) if rng.random() < 0.8 else None
            tb = rnd(N, Cout) if extra == "temb" else None
            rs = rnd(N, OH, OW, Cout) if extra == "res" else None
            outs = []
            for big in (0, 1):
                assert lib.dm_set_option(b"igemm_big", big) == 0
                outs.append(U.op_igemm(x, w, b, X2=x2, temb=tb, res=rs, mode=mode, epi=epi, OH=OH, OW=OW))
            n_pers += lib.dm_op_igemm_tile(N * OH * OW, c_tot, Cout, mode)
            assert not torch.isnan(outs[1].float()).any(), (it, mode, N, H, W, C1, C2, Cout, epi, extra)
            assert torch.equal(outs[0], outs[1]), (it, mode, N, H, W, C1, C2, Cout, epi, extra)
    finally:
        lib.dm_set_option(b"igemm_big", -1)
    assert n_pers >= 20


def test_persistent_tile_launches_overlapping_on_two_streams():
    """Two streams each issue a chain of persistent-tile launches with no synchronisation between the streams, so
    launches overlap on the device (each is 40-320 tiles, well under a full chip).  Every launch takes its own set of
    tile counters, so each result equals the serial one bit for bit."""
    d = U.dev()
    g = torch.Generator(device="cuda").manual_seed(5)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=d, dtype=torch.float32) * scale).half()
    from diff_mining_amd import engine as E
    lib = E.load_library()
    cases = []
    for M, K, Cout in [(10240, 640, 640), (40960, 320, 320), (20480, 1280, 1280), (81920, 320, 640)]:
        cases.append((rnd(1, 1, M, K), rnd(Cout, K, scale=K ** -0.5), rnd(Cout, scale=0.1)))
    try:
        assert lib.dm_set_option(b"igemm_big", 1) == 0
        for x, w, b in cases:
            assert lib.dm_op_igemm_tile(x.shape[2], x.shape[3], w.shape[0], 0) == 1
        serial = [U.op_igemm(x, w, b) for x, w, b in cases]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        torch.cuda.synchronize()
        outs = []
        for rep in range(6):
            for i, (x, w, b) in enumerate(cases):
                with torch.cuda.stream(streams[(i + rep) % 2]):
                    outs.append((i, U.op_igemm(x, w, b, sync=False)))
        torch.cuda.synchronize()
    finally:
        lib.dm_set_option(b"igemm_big", -1)
    for i, y in outs:
        assert torch.equal(y, serial[i]), i


@pytest.mark.parametrize("case", ["conv_temb", "conv_cat_res", "conv_150_samples", "dense_res_ragged", "ln_plain", "conv_8x8"])
def test_head_tail_split_is_bit_identical(case):
    """640 tiles of 256 x 320 on 256 CUs are 2.5 rounds: launch_igemm keeps the two full rounds on the persistent kernel
    and runs the remaining rows on the 128-row tile (`igemm_tail`).  The cut lies on a tile and sample boundary and both
    kernels give the same bits, so the result equals the single-kernel one; `dm_op_igemm_head_rows` reports the cut."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    g = torch.Generator(device="cuda").manual_seed(17)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=d, dtype=torch.float32) * scale).half()
    N, H, W, C1, C2, Cout, mode, temb, res, ln = {
        "conv_temb": (160, 16, 16, 1280, 0, 1280, 1, True, False, False),
        "conv_cat_res": (160, 16, 16, 640, 640, 1280, 1, False, True, False),
        "conv_150_samples": (150, 16, 16, 640, 0, 1280, 1, True, False, False),
        "dense_res_ragged": (1, 1, 40960 - 100, 1280, 0, 1280, 0, False, True, False),
        "ln_plain": (1, 1, 40960, 1280, 0, 1280, 0, False, False, True),
        "conv_8x8": (640, 8, 8, 640, 0, 1280, 1, True, False, False),
    }[case]
    taps = 9 if mode else 1
    M = N * H * W
    head = lib.dm_op_igemm_head_rows(M, H * W, C1 + C2, Cout, mode)
    if n_cu == 256:
        assert 0 < head < M and head % 256 == 0 and head % (H * W if mode else 256) == 0, head
        assert (head // 256) * (Cout // 320) % n_cu == 0
    x = rnd(N, H, W, C1)
    x2 = rnd(N, H, W, C2, scale=0.5) if C2 else None
    w = rnd(Cout, taps * (C1 + C2), scale=(taps * (C1 + C2)) ** -0.5)
    b = rnd(Cout, scale=0.1)
    tb = rnd(N, Cout) if temb else None
    rs = rnd(N, H, W, Cout) if res else None
    if ln:
        ln_s = w.float().sum(1).contiguous()
        ln_t = (torch.randn(Cout, generator=g, device=d) * 0.1).contiguous()
        stats = torch.empty(M, 2, dtype=torch.float32, device=d)
        assert lib.dm_op_ln_stats(U.stream(), U.ptr(x), M, C1, 1e-5, U.ptr(stats)) == 0

    def run():
        if ln:
            y = torch.full((M, Cout), float("nan"), dtype=torch.float16, device=d)
            assert lib.dm_op_igemm_ln(U.stream(), U.ptr(x), U.ptr(w), U.ptr(ln_s), U.ptr(ln_t), U.ptr(stats), U.ptr(y), M, C1, Cout, 0) == 0
            torch.cuda.synchronize()
            return y
        return U.op_igemm(x, w, b, X2=x2, temb=tb, res=rs, mode=mode)
    try:
        y_split = run()
        assert lib.dm_set_option(b"igemm_tail", 0) == 0
        assert lib.dm_op_igemm_head_rows(M, H * W, C1 + C2, Cout, mode) in (0, M)
        y_one = run()
        assert lib.dm_set_option(b"igemm_big", 0) == 0
        y_small = run()
    finally:
        lib.dm_set_option(b"igemm_tail", 1)
        lib.dm_set_option(b"igemm_big", -1)
    assert not torch.isnan(y_split.float()).any()
    assert torch.equal(y_split, y_one) and torch.equal(y_split, y_small)


@pytest.mark.parametrize("C,rows", [(320, 4096 + 5), (640, 1031), (1280, 257), (768, 77)])
def test_ln_stats_kernels_against_torch(C, rows):
    """Per-row (mean, rstd) of a token matrix: the grouped-lane kernel (C = 320 / 640 / 1280, `ln_stats_g` = 1: G lanes per
    row, 64 / G rows per load) and the row-per-wave kernel (every other width, or `ln_stats_g` = 0) against fp64 torch,
    with a row count that is not a multiple of the rows per wave."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.randn(rows, C, generator=g, device=d) * 1.7 + 0.3).half()
    ref_mean = x.double().mean(1)
    ref_rstd = (x.double().var(1, unbiased=False) + 1e-5).rsqrt()
    try:
        for opt in (1, 0):
            assert lib.dm_set_option(b"ln_stats_g", opt) == 0
            stats = torch.full((rows, 2), float("nan"), dtype=torch.float32, device=d)
            assert lib.dm_op_ln_stats(U.stream(), U.ptr(x), rows, C, 1e-5, U.ptr(stats)) == 0
            torch.cuda.synchronize()
            assert (stats[:, 0].double() - ref_mean).abs().max().item() < 2e-6
            assert ((stats[:, 1].double() - ref_rstd) / ref_rstd).abs().max().item() < 2e-6
    finally:
        lib.dm_set_option(b"ln_stats_g", 1)


def test_igemm_dispatch_fuzz_over_batch_sizes():
    """Whatever the batch size makes of the tile count — all rows on the 128-row tile, all on the persistent 256 x 320
    tile, or full rounds on the persistent tile plus a 128-row tail — a launch returns the bits of the 128-row kernel
    (`igemm_big` = 0): random sample counts at the 16x16 / 32x32 / 8x8 levels, every epilogue the U-Net uses there."""
    import random
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    rng = random.Random(77)
    g = torch.Generator(device="cuda").manual_seed(78)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=d, dtype=torch.float32) * scale).half()
    kinds = {0: 0, 1: 0, 2: 0}
    try:
        for it in range(24):
            H = W = rng.choice([8, 16, 16, 16, 32])
            N = rng.randint(1, 40) if it % 3 == 0 else rng.randint(*{8: (300, 900), 16: (90, 260), 32: (25, 70)}[H])
            Cin, Cout = rng.choice([(640, 1280), (1280, 1280), (640, 640), (1280, 640)])
            mode = rng.choice([0, 1, 1])
            extra = rng.choice(["", "temb", "res"]) if mode else rng.choice(["", "res"])
            taps = 9 if mode else 1
            x = rnd(N, H, W, Cin)
            w = rnd(Cout, taps * Cin, scale=(taps * Cin) ** -0.5)
            b = rnd(Cout, scale=0.1)
            tb = rnd(N, Cout) if extra == "temb" else None
            rs = rnd(N, H, W, Cout) if extra == "res" else None
            M = N * H * W
            assert lib.dm_set_option(b"igemm_big", -1) == 0
            head = lib.dm_op_igemm_head_rows(M, H * W, Cin, Cout, mode)
            kinds[0 if head == 0 else (1 if head == M else 2)] += 1
            y = U.op_igemm(x, w, b, temb=tb, res=rs, mode=mode)
            assert lib.dm_set_option(b"igemm_big", 0) == 0
            y0 = U.op_igemm(x, w, b, temb=tb, res=rs, mode=mode)
            assert torch.equal(y, y0), (it, N, H, Cin, Cout, mode, extra, head)
    finally:
        lib.dm_set_option(b"igemm_big", -1)
    print("dispatch kinds (128-row / persistent / head+tail):", kinds)
    assert kinds[0] > 0 and kinds[1] > 0 and kinds[2] > 0


def test_attention_cross_fuzz():
    """The resident-K/V cross-attention kernel on random (samples, query count, prompt slots) against the generic kernel."""
    import random
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    rng = random.Random(5)
    try:
        for it in range(10):
            D = rng.choice([40, 80])
            heads, Tk = 8, 77
            B = rng.randint(1, 9)
            Tq = rng.choice([256, 300, 1024, 1111, 4096])
            P = rng.randint(1, 3)
            Cc = heads * D
            q = U.f16_randn(B, Tq, Cc, seed=100 + it).to(d)
            kv = U.f16_randn(P, Tk, 2 * Cc, seed=200 + it).to(d)
            slots = torch.tensor([rng.randrange(P) for _ in range(B)], dtype=torch.int32, device=d)
            assert lib.dm_set_option(b"attn_cross", 1) == 0
            o = U.op_attention(q, kv[..., :Cc], kv[..., Cc:], heads, slots=slots)
            assert lib.dm_set_option(b"attn_cross", 0) == 0
            o0 = U.op_attention(q, kv[..., :Cc], kv[..., Cc:], heads, slots=slots)
            assert not torch.isnan(o.float()).any()
            U.assert_close_fp16(o, o0.float().cpu(), f"cross fuzz it={it} D={D} B={B} Tq={Tq}", rel=3e-3, abs_frac=4e-3)
    finally:
        lib.dm_set_option(b"attn_cross", 1)


def test_attention_self_16384_keys_vs_fp32_softmax():
    """The X-ray case (BASELINE configs[4]: 1024 px -> latent 128 x 128 = 16 384 tokens, head_dim 40): the online softmax
    over 256 key tiles against an exact fp32 softmax, on 640 sampled query rows per head (all rows would be a 8.6 GB score
    tensor).  Scores are scaled so that the softmax is neither flat nor one-hot (row max ~ 5 sigma over 16 384 keys)."""
    heads, D, T = 8, 40, 16384
    Cc = heads * D
    qkv = U.f16_randn(1, T, 3 * Cc, seed=41)
    qkv[..., :Cc] *= 2.5                                      # score sigma 2.5: the largest of 16 384 weights per row is ~0.05-0.5
    g = qkv.to(U.dev())
    o = U.op_attention(g[..., :Cc], g[..., Cc:2 * Cc], g[..., 2 * Cc:], heads)
    rows = torch.cat([torch.arange(0, 64), torch.randperm(T, generator=torch.Generator().manual_seed(3))[:512], torch.arange(T - 64, T)]).to(U.dev())
    q = g[0, rows, :Cc].float().view(-1, heads, D).transpose(0, 1)              # [heads, R, D]
    k = g[0, :, Cc:2 * Cc].float().view(T, heads, D).transpose(0, 1)
    v = g[0, :, 2 * Cc:].float().view(T, heads, D).transpose(0, 1)
    p = torch.softmax(q @ k.transpose(1, 2) * D ** -0.5, dim=-1)
    ref = (p @ v).transpose(0, 1).reshape(len(rows), Cc)
    got = o[0, rows].float()
    r, m = U.assert_close_fp16(got, ref, "self-attn D=40 T=16384", rel=3e-3, abs_frac=6e-3)
    print(f"self-attention 16 384 keys, D = 40, {len(rows)} sampled rows x 8 heads: rel-L2 {r:.2e}, max|err|/max|ref| {m:.2e}; "
          f"largest softmax weight {p.max().item():.3f}")
    assert not torch.isnan(o.float()).any()


@pytest.mark.parametrize("M,C,Cout,epi", [(300, 320, 960, 0), (4096 + 3, 640, 1920, 0), (131072 + 6, 320, 2560, 1), (40960, 1280, 10240, 1),
                                          (655360, 320, 320, 0)])
def test_layernorm_statistics_inside_the_gemm(M, C, Cout, epi):
    """r03: LN -> Linear with NO statistics kernel (dm_op_igemm_ln with stats = NULL): K = C, so the GEMM's own k loop carries
    every channel of a row through LDS and accumulates (sum, sum of squares) there.  (1) against F.linear(F.layer_norm(x))
    in fp32 on rows with a non-zero mean (one-pass variance: the cancellation case); (2) the 128-row tile and the persistent
    256 x 320 tile add the same numbers in the same order: bit-identical; (3) against the statistics-kernel path: the two
    differ only by the statistics' rounding (two-pass vs one-pass fp32), far below one fp16 ulp of the output on average."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    x = (U.f16_randn(M, C, seed=1).float() * 1.5 + 2.0 * U.f16_randn(M, 1, seed=2).float()).half()     # |mean| up to ~3 sigma of the row
    w = U.f16_randn(Cout, C, seed=3, scale=C ** -0.5)
    b = U.f16_randn(Cout, seed=4, scale=0.1)
    gamma = (1 + 0.1 * U.f16_randn(C, seed=5).float()).half()
    beta = (0.05 * U.f16_randn(C, seed=6).float()).half()
    wq, bq = U.pack_geglu(w, b) if epi == 1 else (w, b)
    wf = (wq.float() * gamma.float()[None]).half()
    ln_s, ln_t = wf.float().sum(1), (wq.float() @ beta.float()) + bq.float()
    xg, wfd, sd_, td_ = x.to(d), wf.to(d), ln_s.to(d), ln_t.to(d)
    stats = torch.empty(M, 2, dtype=torch.float32, device=d)
    assert lib.dm_op_ln_stats(U.stream(), U.ptr(xg), M, C, 1e-5, U.ptr(stats)) == 0

    def run(st):
        y = torch.full((M, Cout // 2 if epi else Cout), float("nan"), dtype=torch.float16, device=d)
        assert lib.dm_op_igemm_ln(U.stream(), U.ptr(xg), U.ptr(wfd), U.ptr(sd_), U.ptr(td_), U.ptr(st) if st is not None else None,
                                  U.ptr(y), M, C, Cout, epi) == 0
        torch.cuda.synchronize()
        return y
    try:
        lib.dm_set_option(b"igemm_big", 0)
        y_small = run(None)
        lib.dm_set_option(b"igemm_big", 1)
        y_big = run(None)
        y_kernel = run(stats)
    finally:
        lib.dm_set_option(b"igemm_big", -1)
    y_auto = run(None)                                         # per-shape choice incl. the head / tail row split
    assert not torch.isnan(y_big.float()).any()
    assert torch.equal(y_small, y_big) and torch.equal(y_auto, y_big)
    rows = torch.arange(M) if M <= 8192 else torch.cat([torch.arange(0, 2048), torch.arange(M - 2048, M)])
    xr = x[rows].float()
    h = F.linear(F.layer_norm(xr, (C,), gamma.float(), beta.float(), 1e-5), w.float(), b.float())
    if epi == 1:
        hh = h.half().float()
        ref = hh[:, :Cout // 2] * F.gelu(hh[:, Cout // 2:]).half().float()
    else:
        ref = h
    r, m = U.assert_close_fp16(y_big[rows.to(d)], ref, f"in-GEMM LN statistics epi={epi}", rel=3e-3, abs_frac=4e-3)
    diff = (y_big.float() - y_kernel.float()).abs()
    frac = (diff > 0).float().mean().item()
    print(f"in-GEMM LN statistics M={M} C={C} epi={epi}: rel-L2 vs fp32 {r:.2e}; outputs that differ from the statistics-kernel path: {frac:.2%}, "
          f"max |d| {diff.max().item():.2e}")
    assert frac < 0.05 and diff.max().item() <= 4e-3 * ref.abs().max().item()


@pytest.mark.parametrize("ratio", [10.0, 30.0, 100.0])
def test_layernorm_fold_at_large_mean_over_std(ratio):
    """rows whose |mean| is `ratio` x their spread — where a one-pass variance E[x^2] - mean^2 cancels.  The folded
    LN -> Linear with the statistics inside the GEMM (the default up to N = 2560) and with the two-pass statistics kernel, both
    against F.linear(F.layer_norm(x)) in fp32.  x is fp16, so at ratio 100 the INPUT itself carries the row's spread with
    only ~4 significant bits more than the mean's ulp: the reference LayerNorm output is what it is; the question is only
    whether the folded forms add to its error."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    M, C, Cout = 4096, 320, 960
    sign = torch.where(torch.arange(M) % 2 == 0, 1.0, -1.0)[:, None]
    x = (U.f16_randn(M, C, seed=11).float() * 0.5 + sign * 0.5 * ratio).half()
    got_ratio = (x.float().mean(1).abs() / x.float().std(1)).median().item()
    w = U.f16_randn(Cout, C, seed=3, scale=C ** -0.5)
    b = U.f16_randn(Cout, seed=4, scale=0.1)
    gamma = (1 + 0.1 * U.f16_randn(C, seed=5).float()).half()
    beta = (0.05 * U.f16_randn(C, seed=6).float()).half()
    wf = (w.float() * gamma.float()[None]).half()
    ln_s, ln_t = wf.float().sum(1), (w.float() @ beta.float()) + b.float()
    xg, wfd, sd_, td_ = x.to(d), wf.to(d), ln_s.to(d), ln_t.to(d)
    stats = torch.empty(M, 2, dtype=torch.float32, device=d)
    assert lib.dm_op_ln_stats(U.stream(), U.ptr(xg), M, C, 1e-5, U.ptr(stats)) == 0

    def run(st):
        y = torch.full((M, Cout), float("nan"), dtype=torch.float16, device=d)
        assert lib.dm_op_igemm_ln(U.stream(), U.ptr(xg), U.ptr(wfd), U.ptr(sd_), U.ptr(td_), U.ptr(st) if st is not None else None,
                                  U.ptr(y), M, C, Cout, 0) == 0
        torch.cuda.synchronize()
        return y.float().cpu()
    y_in, y_st = run(None), run(stats)
    xr = x.double()
    ref = F.linear(F.layer_norm(xr, (C,), gamma.double(), beta.double(), 1e-5), w.double(), b.double()).float()
    unf = F.linear(F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5).half().float(), w.float(), b.float()).half().float()   # the unfused pair's roundings
    rel = lambda a: ((a - ref).norm() / ref.norm()).item()        # noqa: E731
    mu = stats.cpu()[:, 0]
    print(f"LN fold at |mean|/std = {got_ratio:.0f}: rel-L2 vs fp64 LayerNorm+Linear: statistics in the GEMM {rel(y_in):.2e}, statistics kernel "
          f"{rel(y_st):.2e}, unfused fp16 LN output -> Linear (the autocast pair) {rel(unf):.2e}; max |mean err| of the kernel {(mu - x.float().mean(1)).abs().max():.1e}")
    assert rel(y_st) <= max(1.5 * rel(unf), 6e-4)
    assert rel(y_in) <= max(1.5 * rel(y_st), 6e-4), "one-pass statistics inside the GEMM lose precision at this |mean|/std"


@pytest.mark.parametrize("N,H,W,C,Cout", [(3, 16, 16, 320, 320), (2, 32, 32, 640, 640), (160, 64, 64, 320, 320), (5, 16, 8, 64, 160)])
def test_groupnorm_folded_into_conv1x1(N, H, W, C, Cout):
    """r03: `Transformer2DModel.norm` (GroupNorm 32, eps 1e-6, no activation) folded into `proj_in` (1x1 convolution): per-sample
    weights fp16(W diag(a_n)) + fp32 bias rows, the GEMM on the raw tensor (dm_op_groupnorm_conv1x1) vs
    F.conv2d(F.group_norm(x)) in fp32; and the persistent 256 x 320 tile against the 128-row tile: bit-identical."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    G = 32
    g = torch.Generator().manual_seed(C + N)
    x = (torch.randn(N, C, H, W, generator=g) * (1.0 + torch.rand(N, C, 1, 1, generator=g)) + torch.randn(N, C, 1, 1, generator=g)).half()
    w = U.f16_randn(Cout, C, seed=2, scale=C ** -0.5)
    b = U.f16_randn(Cout, seed=3, scale=0.1)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).float()
    beta = (0.1 * torch.randn(C, generator=g)).float()
    ref = F.conv2d(F.group_norm(x.float(), G, gamma, beta, 1e-6), w.float()[:, :, None, None], b.float())
    xg = U.to_nhwc(x).to(d)
    wg, bg, gg, be = w.to(d), b.to(d), gamma.to(d), beta.to(d)

    def run():
        y = torch.full((N, H, W, Cout), float("nan"), dtype=torch.float16, device=d)
        assert lib.dm_op_groupnorm_conv1x1(U.stream(), U.ptr(xg), N, H * W, C, G, 1e-6, U.ptr(gg), U.ptr(be), U.ptr(wg), U.ptr(bg), Cout, U.ptr(y)) == 0
        torch.cuda.synchronize()
        return y
    try:
        lib.dm_set_option(b"igemm_big", 0)
        y0 = run()
        lib.dm_set_option(b"igemm_big", 1)
        y1 = run() if (Cout % 320 == 0 and (H * W) % 256 == 0) else y0
    finally:
        lib.dm_set_option(b"igemm_big", -1)
    ya = run()
    assert not torch.isnan(y0.float()).any()
    assert torch.equal(y0, y1) and torch.equal(ya, y0)
    sel = slice(None) if N <= 8 else [0, N // 2, N - 1]
    r, m = U.assert_close_fp16(U.to_nchw(y0[sel].cpu()), ref[sel], "GroupNorm folded into conv1x1", rel=3e-3, abs_frac=4e-3)
    # the unfused pair (GroupNorm kernel -> fp16 -> igemm) on the same data, for scale
    yn = U.op_groupnorm(xg, gg, be, G, 1e-6, False)
    yu = U.op_igemm(yn, wg, bg)
    r_u = U.rel_l2(U.to_nchw(yu[sel].cpu()), ref[sel])
    print(f"GN folded into conv1x1 N={N} {H}x{W} C={C}: rel-L2 vs fp32 {r:.2e} (unfused pair: {r_u:.2e}), max|err|/max|ref| {m:.2e}")


@pytest.mark.parametrize("N,H,W,C3,C4,Cout", [(2, 9, 7, 128, 64, 320), (160, 32, 32, 1280, 640, 640), (161, 16, 16, 640, 0, 1280), (3, 16, 16, 320, 0, 640)])
def test_conv_shortcut_folded_into_conv2(N, H, W, C3, C4, Cout):
    """r03: `ResnetBlock2D` output = conv_shortcut(x) + conv2(h), with the 1x1 shortcut on the (possibly concatenated) block
    input folded into conv2's k loop (dm_op_igemm_shortcut: weight rows [9 * Cout (tap, c) | C3 + C4], bias = b2 + b_sc), against
    F.conv2d + F.conv2d in fp32 on sampled images; the persistent 256 x 320 tile against the 128-row tile: bit-identical (ragged
    last tile included); and against the unfused pair (shortcut GEMM -> fp16 -> residual of conv2)."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    g = torch.Generator(device="cuda").manual_seed(5)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=d, dtype=torch.float32) * scale).half()
    Cin = Cout                                          # conv2: Cout -> Cout
    h = rnd(N, H, W, Cin)
    x3 = rnd(N, H, W, C3)
    x4 = rnd(N, H, W, C4, scale=0.5) if C4 else None
    w2 = U.f16_randn(Cout, Cin, 3, 3, seed=6, scale=(9 * Cin) ** -0.5)
    ws = U.f16_randn(Cout, C3 + C4, seed=7, scale=(C3 + C4) ** -0.5)
    b2, bs = U.f16_randn(Cout, seed=8, scale=0.1), U.f16_randn(Cout, seed=9, scale=0.1)
    wp = torch.cat([U.pack_conv3(w2), ws], dim=1).contiguous().to(d)
    bias = (b2.float() + bs.float()).half().to(d)

    def run():
        y = torch.full((N, H, W, Cout), float("nan"), dtype=torch.float16, device=d)
        assert lib.dm_op_igemm_shortcut(U.stream(), U.ptr(h), U.ptr(x3), U.ptr(x4), U.ptr(wp), U.ptr(bias), None, U.ptr(y), N, H, W, Cin, C3, C4, Cout, 1) == 0
        torch.cuda.synchronize()
        return y
    try:
        lib.dm_set_option(b"igemm_big", 0)
        y0 = run()
        lib.dm_set_option(b"igemm_big", 1)
        y1 = run() if Cout % 320 == 0 else y0
    finally:
        lib.dm_set_option(b"igemm_big", -1)
    ya = run()
    assert not torch.isnan(y0.float()).any()
    assert torch.equal(y0, y1) and torch.equal(ya, y0)
    for n in sorted({0, N // 2, N - 1}):
        xin = x3[n:n + 1] if x4 is None else torch.cat([x3[n:n + 1], x4[n:n + 1]], 3)
        ref = F.conv2d(U.to_nchw(h[n:n + 1].cpu().float()), w2.float(), b2.float(), padding=1) + \
            F.conv2d(U.to_nchw(xin.cpu().float()), ws.float()[:, :, None, None], bs.float())
        r, m = U.assert_close_fp16(U.to_nchw(y0[n:n + 1].cpu()), ref, f"conv2 + folded shortcut n={n}")
    # the unfused pair on the same data
    sc = U.op_igemm(x3, ws.to(d), bs.to(d), X2=x4, mode=0)
    yu = U.op_igemm(h, U.pack_conv3(w2).to(d), b2.to(d), res=sc, mode=1)
    diff = (yu.float() - y0.float()).abs()
    print(f"conv2 + folded shortcut N={N} {H}x{W} {C3}+{C4}->{Cout}: rel-L2 vs fp32 {r:.2e}; vs the unfused pair: {(diff > 0).float().mean().item():.1%} of the outputs differ, "
          f"max |d| {diff.max().item():.2e}")
    assert diff.max().item() <= 4e-3 * y0.float().abs().max().item()


@pytest.mark.parametrize("M,C", [(4096 + 5, 320), (163840, 640), (40960, 1280)])
def test_ff2_residual_proj_out_as_one_gemm(M, C):
    """r03: out = proj_out(ff.net.2(f) + t2) + x is a linear chain, = (Wp W2) f + Wp t2 + (Wp b2 + bp) + x: ONE GEMM over [f | t2]
    with the pre-multiplied weights (dm_op_igemm_shortcut, dense mode, residual x) against the two F.linear calls in fp32, and
    against the unfused pair of GEMMs (which rounds the intermediate to fp16); both tiles bit-identical."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    g = torch.Generator(device="cuda").manual_seed(9)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=d, dtype=torch.float32) * scale).half()
    f, t2, x = rnd(M, 4 * C, scale=0.7), rnd(M, C), rnd(M, C)
    w2 = rnd(C, 4 * C, scale=(4 * C) ** -0.5)
    wp = rnd(C, C, scale=C ** -0.5)
    b2, bp = rnd(C, scale=0.1), rnd(C, scale=0.1)
    wm = torch.cat([(wp.float() @ w2.float()).half(), wp], dim=1).contiguous()
    bm = (wp.float() @ b2.float() + bp.float()).half()

    def run():
        y = torch.full((M, C), float("nan"), dtype=torch.float16, device=d)
        assert lib.dm_op_igemm_shortcut(U.stream(), U.ptr(f), U.ptr(t2), None, U.ptr(wm), U.ptr(bm), U.ptr(x), U.ptr(y), 1, 1, M, 4 * C, C, 0, C, 0) == 0
        torch.cuda.synchronize()
        return y
    try:
        lib.dm_set_option(b"igemm_big", 0)
        y0 = run()
        lib.dm_set_option(b"igemm_big", 1)
        y1 = run() if C % 320 == 0 else y0
    finally:
        lib.dm_set_option(b"igemm_big", -1)
    ya = run()
    assert not torch.isnan(y0.float()).any()
    assert torch.equal(y0, y1) and torch.equal(ya, y0)
    rows = torch.arange(M, device=d) if M <= 8192 else torch.cat([torch.arange(0, 2048, device=d), torch.arange(M - 2048, M, device=d)])
    t3 = F.linear(f[rows].float(), w2.float(), b2.float()) + t2[rows].float()
    ref = F.linear(t3, wp.float(), bp.float()) + x[rows].float()
    r, m = U.assert_close_fp16(y0[rows], ref, "ff2 + residual + proj_out as one GEMM", rel=3e-3, abs_frac=4e-3)
    t3u = U.op_igemm(f.view(1, 1, M, 4 * C), w2, b2, res=t2.view(1, 1, M, C), mode=0)
    yu = U.op_igemm(t3u, wp, bp, res=x.view(1, 1, M, C), mode=0).view(M, C)
    ru = U.rel_l2(yu[rows], ref)
    print(f"ff2 + proj_out as one GEMM M={M} C={C}: rel-L2 vs fp32 {r:.2e} (the unfused pair: {ru:.2e}), max|err|/max|ref| {m:.2e}")


@pytest.mark.parametrize("W,C1,C2,Cout,N", [(128, 320, 0, 320, 2), (128, 320, 320, 320, 1), (64, 320, 0, 320, 6), (64, 640, 320, 320, 3), (32, 640, 0, 640, 10), (32, 1280, 640, 640, 6),
                                           (16, 1280, 0, 1280, 24), (16, 640, 0, 1280, 17)])
def test_igemm_tap_reuse_tile(W, C1, C2, Cout, N):
    """igemm_pers_tr.hip (option tap_reuse; 2 = every eligible width): the 3x3 stride-1 convolutions with the k order (dy, slab, dx) and ONE activation
    stage per three horizontal taps, against F.conv2d in fp32 (plain, + time embedding, + residual, channel concat), and bit for
    bit against the 128-row tile's KO variant (igemm_ko.hip) on the same operands — the pair that makes a sample's result
    independent of the batch it rides in."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    H = W
    Cin = C1 + C2
    g = torch.Generator(device="cuda").manual_seed(51)
    x1 = torch.randn(N, H, W, C1, generator=g, device=d, dtype=torch.float32).half()
    x2 = (torch.randn(N, H, W, C2, generator=g, device=d, dtype=torch.float32) * 0.5).half() if C2 else None
    w = U.f16_randn(Cout, Cin, 3, 3, seed=52, scale=(9 * Cin) ** -0.5)
    b = U.f16_randn(Cout, seed=53, scale=0.1)
    temb = U.f16_randn(N, Cout, seed=54)
    res = torch.randn(N, H, W, Cout, generator=g, device=d, dtype=torch.float32).half()
    wg, bg = U.pack_conv3(w).to(d), b.to(d)
    out = {}
    try:
        assert lib.dm_set_option(b"tap_reuse", 2) == 0
        for big in (1, 0):
            assert lib.dm_set_option(b"igemm_big", big) == 0
            out[big] = (U.op_igemm(x1, wg, bg, X2=x2, mode=1), U.op_igemm(x1, wg, bg, X2=x2, temb=temb.to(d), mode=1),
                        U.op_igemm(x1, wg, bg, X2=x2, res=res, mode=1))
    finally:
        lib.dm_set_option(b"igemm_big", -1)
        lib.dm_set_option(b"tap_reuse", 1)
    for k, name in enumerate(("plain", "temb", "res")):
        assert torch.equal(out[1][k], out[0][k]), f"tap-reuse tile != 128-row KO tile ({name}): " \
            f"{(out[1][k].float() - out[0][k].float()).abs().max().item():.3e}"
    y_plain, y_temb, y_res = out[1]
    for n in sorted({0, N // 2, N - 1}):
        xin = x1[n:n + 1] if x2 is None else torch.cat([x1[n:n + 1], x2[n:n + 1]], 3)
        ref = F.conv2d(U.to_nchw(xin.cpu().float()), w.float(), b.float(), padding=1)
        U.assert_close_fp16(U.to_nchw(y_plain[n:n + 1]), ref, f"tap-reuse conv3x3 n={n}")
        U.assert_close_fp16(U.to_nchw(y_temb[n:n + 1]), ref.half().float() + temb[n].float()[None, :, None, None], f"tap-reuse +temb n={n}")
        U.assert_close_fp16(U.to_nchw(y_res[n:n + 1]), ref.half().float() + U.to_nchw(res[n:n + 1].cpu().float()), f"tap-reuse +res n={n}")


def test_igemm_tap_reuse_upsample():
    """The tap-reuse tile on Upsample2D.conv (nearest 2x, then 3x3) onto a 64-pixel-wide image: against F.conv2d on the
    F.interpolate'd input in fp32, and bit for bit against the 128-row tile's KO variant."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    N, H, W, C, Cout = 5, 32, 32, 640, 640
    g = torch.Generator(device="cuda").manual_seed(61)
    x = torch.randn(N, H, W, C, generator=g, device=d, dtype=torch.float32).half()
    w = U.f16_randn(Cout, C, 3, 3, seed=62, scale=(9 * C) ** -0.5)
    b = U.f16_randn(Cout, seed=63, scale=0.1)
    wg, bg = U.pack_conv3(w).to(d), b.to(d)
    out = {}
    try:
        assert lib.dm_set_option(b"tap_reuse", 2) == 0          # (the up-sampling layer is not in the default rule: +3 %)
        for big in (1, 0):
            assert lib.dm_set_option(b"igemm_big", big) == 0
            out[big] = U.op_igemm(x, wg, bg, mode=3, OH=2 * H, OW=2 * W)
        assert lib.dm_set_option(b"tap_reuse", 0) == 0
        std = U.op_igemm(x, wg, bg, mode=3, OH=2 * H, OW=2 * W)
    finally:
        lib.dm_set_option(b"igemm_big", -1)
        lib.dm_set_option(b"tap_reuse", 1)
    assert torch.equal(out[1], out[0]), f"tap-reuse up-sampling tile != 128-row KO tile: {(out[1].float() - out[0].float()).abs().max().item():.3e}"
    for n in (0, N - 1):
        up = F.interpolate(U.to_nchw(x[n:n + 1].cpu().float()), scale_factor=2, mode="nearest")
        ref = F.conv2d(up, w.float(), b.float(), padding=1)
        U.assert_close_fp16(U.to_nchw(out[1][n:n + 1]), ref, f"tap-reuse upsample conv n={n}")
    assert (std.float() - out[1].float()).abs().max().item() <= 2e-3 * std.float().abs().max().item()


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 320, 320), (3, 5, 7, 128, 320), (3, 12, 10, 64, 640), (5, 32, 32, 640, 640),
                                            (1, 1, 2, 64, 320)])
def test_upconv_folded(N, H, W, Cin, Cout):
    """igemm_pers_up.hip (option up_fold): diffusers' Upsample2D — F.interpolate(scale_factor=2, nearest) then Conv2d 3x3 pad 1 — as
    four 2x2 convolutions on the source grid.  (1) the host weight fold against a torch construction, exactly; (2) the kernel
    against F.conv2d on the F.interpolate'd input in fp32 (the layer's definition, original fp16 weights); (3) against the same
    2x2 convolutions evaluated by torch in fp32 with the folded fp16 weights (the kernel's own arithmetic: fp32-accumulation
    distance); (4) against the unfolded kernel (mode 3).  Ragged shapes: rows per parity class that are not a multiple of 256, a
    single-row image, one tile shared by all parity classes' tails."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    g = torch.Generator(device="cuda").manual_seed(71)
    x = torch.randn(N, H, W, Cin, generator=g, device=d, dtype=torch.float32).half()
    w = U.f16_randn(Cout, Cin, 3, 3, seed=72, scale=(9 * Cin) ** -0.5)
    b = U.f16_randn(Cout, seed=73, scale=0.1)
    w4_ref = U.fold_upconv_torch(w)
    w4 = torch.empty(4, Cout, 4 * Cin, dtype=torch.float16)
    assert lib.dm_op_fold_upconv_weights(U.C.c_void_p(w.contiguous().data_ptr()), Cout, Cin, U.C.c_void_p(w4.data_ptr())) == 0
    assert torch.equal(w4, w4_ref), "host weight fold differs from the torch construction"
    y = U.op_upconv_folded(x, w4.to(d), b.to(d))
    std = U.op_igemm(x, U.pack_conv3(w).to(d), b.to(d), mode=3, OH=2 * H, OW=2 * W)
    xs = U.to_nchw(x.cpu().float())
    ref = F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), w.float(), b.float(), padding=1)
    r, m = U.assert_close_fp16(U.to_nchw(y), ref, "folded upsample conv vs interpolate + conv2d")
    r0, m0 = U.assert_close_fp16(U.to_nchw(std), ref, "unfolded upsample conv vs interpolate + conv2d")
    # the fold's own arithmetic in fp32: per parity class a 2x2 convolution of the padded source with the folded fp16 weights
    own = torch.empty_like(ref)
    for py in (0, 1):
        for px in (0, 1):
            k = w4[py * 2 + px].float().view(Cout, 2, 2, Cin).permute(0, 3, 1, 2)          # [Cout, Cin, a, b]
            xp = F.pad(xs, (1 - px, px, 1 - py, py))                                       # tap (a, b) reads (y - 1 + py + a, x - 1 + px + b)
            own[:, :, py::2, px::2] = F.conv2d(xp, k, b.float())
    U.assert_close_fp16(U.to_nchw(y), own, "folded upsample conv vs its own arithmetic in fp32", rel=4e-4, abs_frac=1.2e-3)
    d_fold = U.rel_l2(U.to_nchw(y), U.to_nchw(std))
    print(f"upconv fold N={N} {H}x{W} {Cin}->{Cout}: vs definition rel-L2 {r:.2e} (unfolded kernel {r0:.2e}); folded vs unfolded {d_fold:.2e}")
    # the one extra rounding: a summed tap is 2 or 4 fp16 weights rounded to fp16 once, an independent 2^-12-relative error per
    # folded weight, i.e. about one more fp16 rounding of the output (measured: 2.1e-4 -> 2.9e-4 against the definition, per layer;
    # end to end the distance to the fp32 oracle moves 8.99e-4 -> 9.11e-4: tests/test_gpu_e2e.py::test_up_fold_option_end_to_end)
    assert r <= 1.6 * r0 + 1e-5 and r < 4e-4, (r, r0)
    assert d_fold < 6e-4


def test_upconv_folded_refuses_what_it_does_not_take():
    """The folded up-sampler exists on the 320-channel persistent tile only: other shapes must fail loudly (nonzero return, nothing
    written), never fall back silently — the engine keeps the unfolded layer for them (igemm_up4_ok)."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    x = torch.zeros(1, 4, 4, 64, dtype=torch.float16, device=d)
    y = torch.full((1, 8, 8, 320), 7.0, dtype=torch.float16, device=d)
    w4 = torch.zeros(4, 320, 4 * 64, dtype=torch.float16, device=d)
    b = torch.zeros(320, dtype=torch.float16, device=d)
    args = (U.stream(), U.ptr(x), U.ptr(w4), U.ptr(b), U.ptr(y))
    assert lib.dm_op_upconv_folded(*args, 1, 4, 4, 64, 160) != 0          # Cout not a multiple of 320
    assert lib.dm_op_upconv_folded(*args, 1, 4, 4, 48, 320) != 0          # Cin not a multiple of 64
    assert lib.dm_op_upconv_folded(*args, 1, 600, 4, 64, 320) != 0        # packed row coordinates: side <= 511
    assert lib.dm_op_upconv_folded(*args, 1, 1, 1, 64, 320) != 0          # a single source row per class
    torch.cuda.synchronize()
    assert (y == 7.0).all()
    assert lib.dm_op_fold_upconv_weights(None, 320, 64, None) != 0


def _block_sums_torch(y2d):
    """[rows][C] fp16 -> [rows / 64][C] fp64: (sum, sum of squares) of every channel pair over 64-row blocks, exact in fp64."""
    rows, C = y2d.shape
    v = y2d.double().view(rows // 64, 64, C // 2, 2)
    s = v.sum(dim=(1, 3))
    q = (v * v).sum(dim=(1, 3))
    return torch.stack([s, q], dim=-1).reshape(rows // 64, C)


@pytest.mark.parametrize("N,H,W,Cin,Cout,tap_reuse,expect", [
    (32, 64, 64, 320, 320, 1, "all"),       # the tap-reuse kernel, 64-pixel-wide images (two whole rounds of tiles): every block from the epilogue
    (32, 64, 64, 320, 320, 0, "all"),       # the plain persistent kernel
    (40, 64, 64, 320, 320, 1, "head"),      # 2.5 rounds: the head from the epilogue, the tail rows (128-row tile) from the output
    (160, 16, 16, 1280, 1280, 1, "head"),   # 2.5 rounds of 256-row tiles: the head from the epilogue, the tail rows from the output
    (96, 32, 32, 640, 640, 1, "none"),      # the 32-pixel-wide tap-reuse kernel does not emit them
    (2, 16, 16, 320, 320, 1, "none"),       # a launch the 128-row tile takes
])
def test_groupnorm_block_sums_from_the_conv1_epilogue(N, H, W, Cin, Cout, tap_reuse, expect):
    """r05: norm2's statistics as per-(64-row block, channel pair) fp32 sums.  The persistent kernels write them from conv1's epilogue
    (dm_op_conv_temb_gn_blocks); dm_op_gn_blocks computes them from the tensor.  Both must give the SAME BITS (which of the two produced a
    block depends on the batch size through the tile choice), the conv output must not change, and the sums must be the fp64 sums of
    the fp16 output to fp32 accuracy."""
    import ctypes
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    x = U.f16_randn(N, H, W, Cin, seed=41, scale=0.7).to(d)
    w = U.f16_randn(Cout, 9 * Cin, seed=42, scale=(9 * Cin) ** -0.5).to(d)
    b = U.f16_randn(Cout, seed=43, scale=0.2).to(d)
    temb = U.f16_randn(N, Cout, seed=44, scale=0.5).to(d)
    M = N * H * W
    try:
        lib.dm_set_option(b"tap_reuse", tap_reuse)              # (the option also selects the k order: the reference output under the same one)
        y_ref = U.op_igemm(x, w, bias=b, temb=temb, mode=1)
        y = torch.empty(N, H, W, Cout, dtype=torch.float16, device=d)
        blocks = torch.full((M // 64, Cout), float("nan"), dtype=torch.float32, device=d)
        done = ctypes.c_int(-1)
        assert lib.dm_op_conv_temb_gn_blocks(U.stream(), U.ptr(x), U.ptr(w), U.ptr(b), U.ptr(temb), U.ptr(y), N, H, W, Cin, Cout, Cout,
                                             U.ptr(blocks), ctypes.byref(done)) == 0
        torch.cuda.synchronize()
    finally:
        lib.dm_set_option(b"tap_reuse", 1)
    assert torch.equal(y, y_ref)
    rows_done = done.value
    if expect == "all":
        assert rows_done == M
    elif expect == "none":
        assert rows_done == 0
    else:
        assert 0 < rows_done < M and rows_done % 256 == 0
    assert torch.isnan(blocks[rows_done // 64:]).all()                # nothing written beyond what was promised
    full = torch.full_like(blocks, float("nan"))
    assert lib.dm_op_gn_blocks(U.stream(), U.ptr(y), M, Cout, 0, U.ptr(full)) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(full).all()
    assert torch.equal(blocks[:rows_done // 64], full[:rows_done // 64])      # bit-identical between the two producers
    # the caller's completion: rows [rows_done, M) from the tensor
    assert lib.dm_op_gn_blocks(U.stream(), U.ptr(y), M, Cout, rows_done, U.ptr(blocks)) == 0
    torch.cuda.synchronize()
    assert torch.equal(blocks, full)
    ref = _block_sums_torch(y.view(M, Cout))
    scale = _block_sums_torch(y.view(M, Cout).abs())
    assert ((full.double() - ref).abs() <= 4e-6 * scale + 1e-6).all()
    # statistics -> GroupNorm + SiLU vs F.group_norm in fp32 on the same fp16 tensor, at the tolerance of test_groupnorm
    g = (torch.randn(Cout, generator=torch.Generator().manual_seed(45)) * 0.1 + 1).to(d)
    be = (torch.randn(Cout, generator=torch.Generator().manual_seed(46)) * 0.1).to(d)
    n_chk = min(N, 4)
    out = torch.empty(n_chk, H, W, Cout, dtype=torch.float16, device=d)
    assert lib.dm_op_groupnorm_blocks(U.stream(), U.ptr(y), U.ptr(full), n_chk, H * W, Cout, 32, 1e-5, U.ptr(g), U.ptr(be), 1, U.ptr(out)) == 0
    torch.cuda.synchronize()
    refn = F.silu(F.group_norm(U.to_nchw(y[:n_chk]).float(), 32, g, be, 1e-5))
    U.assert_close_fp16(U.to_nchw(out), refn, "groupnorm from block sums")
    # and the same bits as the r04 statistics pass would give after the apply?  Not required (another summation order) — but a sample's
    # result must not depend on the batch: sample 1 alone
    if N >= 2:
        one = torch.empty(1, H, W, Cout, dtype=torch.float16, device=d)
        bl1 = torch.empty(H * W // 64, Cout, dtype=torch.float32, device=d)
        y1 = y[1:2].contiguous()
        assert lib.dm_op_gn_blocks(U.stream(), U.ptr(y1), H * W, Cout, 0, U.ptr(bl1)) == 0
        assert lib.dm_op_groupnorm_blocks(U.stream(), U.ptr(y1), U.ptr(bl1), 1, H * W, Cout, 32, 1e-5, U.ptr(g), U.ptr(be), 1, U.ptr(one)) == 0
        torch.cuda.synchronize()
        assert torch.equal(one[0], out[1])


@pytest.mark.parametrize("B,H,W", [(3, 64, 64), (2, 12, 10), (2, 16, 24), (1, 128, 128), (2, 8, 8), (1, 64, 85), (2, 5, 3), (1, 4, 170)])
def test_conv_out_with_rows_staged_in_lds(B, H, W):
    """conv_out + eps-MSE (the last two steps of SD.compute_loss, compute.py:100-101): the r05 kernel (conv_out.hip: a strip's input
    rows staged once in LDS, nine taps on the matrix cores) and the per-pixel gather kernel (misc.hip) against F.conv2d in fp64 on the
    same fp16 operands: pred within one fp16 rounding of the exact value, loss = (float(pred) - eps)^2 bit for bit, every strip
    geometry (ragged last strip, W not a multiple of 16, one-row strips at W = 128, W = 170: the geometry the new kernel leaves
    to the old one), and a sample's bits independent of the batch."""
    from diff_mining_amd import engine as E
    lib = E.load_library()
    d = U.dev()
    C0 = 320
    x = U.f16_randn(B, H, W, C0, seed=51, scale=1.0)
    w4 = U.f16_randn(4, C0, 3, 3, seed=52, scale=(9 * C0) ** -0.5)
    bias = U.f16_randn(4, seed=53, scale=0.1)
    eps = torch.randn(B, 4, H, W, generator=torch.Generator().manual_seed(54))
    ref = F.conv2d(U.to_nchw(x).double(), w4.double(), bias.double(), padding=1)
    xd, wd, bd, ed = x.to(d), U.pack_conv3(w4).to(d), bias.to(d), eps.to(d)
    out = {}
    try:
        for v in (1, 0):
            assert lib.dm_set_option(b"conv_out_rows", v) == 0
            loss = torch.full((B, 4, H, W), float("nan"), device=d)
            pred = torch.full((B, 4, H, W), float("nan"), dtype=torch.float16, device=d)
            assert lib.dm_op_conv_out(U.stream(), U.ptr(xd), U.ptr(wd), U.ptr(bd), U.ptr(ed), B, H, W, C0, U.ptr(loss), U.ptr(pred)) == 0
            torch.cuda.synchronize()
            out[v] = (loss, pred)
            err = (pred.double().cpu() - ref).abs()
            ulp = torch.maximum(ref.abs(), torch.tensor(2.0 ** -14, dtype=torch.float64)) * 2.0 ** -10
            assert (err <= 0.5 * ulp + 1e-5).all(), (v, (err / ulp).max().item())       # fp16 rounding of an fp32 sum of 2880 exact products
            assert torch.equal(loss, (pred.float() - ed) ** 2)
        assert lib.dm_set_option(b"conv_out_rows", 1) == 0
        if B > 1:
            l1 = torch.empty(1, 4, H, W, device=d)
            p1 = torch.empty(1, 4, H, W, dtype=torch.float16, device=d)
            x1, e1 = xd[1:2].contiguous(), ed[1:2].contiguous()
            assert lib.dm_op_conv_out(U.stream(), U.ptr(x1), U.ptr(wd), U.ptr(bd), U.ptr(e1), 1, H, W, C0, U.ptr(l1), U.ptr(p1)) == 0
            torch.cuda.synchronize()
            assert torch.equal(p1[0], out[1][1][1]) and torch.equal(l1[0], out[1][0][1])
    finally:
        lib.dm_set_option(b"conv_out_rows", 1)
    same = (out[0][1] == out[1][1]).float().mean().item()
    print(f"conv_out {B}x{H}x{W}: {100 * same:.2f} % of the fp16 outputs equal between the two kernels")
    assert same > 0.98
