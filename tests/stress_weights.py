"""Synthetic SDv1.5 weights OFF the benign operating point, calibrated with the oracle (test infrastructure: imports oracle/).

`synth.synth_state_dict(mode="stress", stress_bias=0)` gives the analytic part (output-channel scales log-uniform over two
decades, x50 outlier channels on the layers whose input is normalised, norm gammas over a decade).  What cannot be set
analytically is the thing VERDICT r03 #3 asks for — |mean| / std of 5 ... 20 at the GroupNorm / LayerNorm inputs — because the
spread of a normalisation input is the accumulated output of everything before it.  So the offsets are CALIBRATED: the fp32
oracle runs at 8x8 with an observer on every normalisation input, and the bias of the layer that wrote that input is set to
    sign * r_site * sigma_site * (1 + 0.05 z_c),      r_site log-uniform in [5, 20] from the integer hash,
sigma_site = the median over groups (tokens) of the input's spread with the current offset removed.  Three sweeps in forward
order settle it (an offset moves everything downstream).  Deterministic: same hash, same oracle, same 8x8 calibration input.

    sites and their writers
      <resnet>.norm2                         <- <resnet>.conv1.bias
      <attn>.transformer_blocks.0.norm1/2/3  <- <attn>.proj_in.bias / attn1.to_out.0.bias / attn2.to_out.0.bias
      <resnet>.norm1, <attn>.norm, conv_norm_out (the residual stream)  <- the bias of the layer that wrote the stream tensor
                                               (conv_in, the previous resnet's conv2, the previous transformer's proj_out,
                                               a sampler's conv); in the up blocks the concatenated skip half keeps the
                                               offset it got when it was written
"""
from __future__ import annotations

import numpy as np
import torch

from diff_mining_amd import synth
from oracle import unet_ref as R

SWEEPS = 1
R_LO, R_HI = 5.0, 20.0


def _stream_writers(cfg=R.SD15_REF):
    """norm site -> bias tensor of the layer that wrote its input, for the residual-stream sites (mirrors unet_forward)."""
    out = {}
    last = "conv_in.bias"
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            out[f"down_blocks.{i}.resnets.{j}.norm1"] = last
            last = f"down_blocks.{i}.resnets.{j}.conv2.bias"
            if cfg.down_has_attn[i]:
                out[f"down_blocks.{i}.attentions.{j}.norm"] = last
                last = f"down_blocks.{i}.attentions.{j}.proj_out.bias"
        if i != nb - 1:
            last = f"down_blocks.{i}.downsamplers.0.conv.bias"
    out["mid_block.resnets.0.norm1"] = last
    last = "mid_block.resnets.0.conv2.bias"
    out["mid_block.attentions.0.norm"] = last
    last = "mid_block.attentions.0.proj_out.bias"
    out["mid_block.resnets.1.norm1"] = last
    last = "mid_block.resnets.1.conv2.bias"
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            out[f"up_blocks.{i}.resnets.{j}.norm1"] = last
            last = f"up_blocks.{i}.resnets.{j}.conv2.bias"
            if cfg.up_has_attn[i]:
                out[f"up_blocks.{i}.attentions.{j}.norm"] = last
                last = f"up_blocks.{i}.attentions.{j}.proj_out.bias"
        if i != nb - 1:
            last = f"up_blocks.{i}.upsamplers.0.conv.bias"
    out["conv_norm_out"] = last
    return out


def site_writer(site: str, stream) -> str:
    if site in stream:
        return stream[site]
    if site.endswith(".norm2") and "transformer_blocks" not in site:
        return site[: -len("norm2")] + "conv1.bias"
    b = site.rsplit(".", 1)[0]                       # ...transformer_blocks.0
    a = b[: -len(".transformer_blocks.0")]
    return {"norm1": a + ".proj_in.bias", "norm2": b + ".attn1.to_out.0.bias", "norm3": b + ".attn2.to_out.0.bias"}[site.rsplit(".", 1)[1]]


def _calib_inputs(hw=8):
    x, eps, t, c = (torch.from_numpy(a) for a in synth.synth_inputs(1, 1, hw, hw, latent_dtype=np.float32))
    nb, tb = torch.cat([eps] * 2), torch.cat([t] * 2)
    cc = torch.cat([c[0:1], c[1:2]]).float()
    return R.add_noise(x.expand(2, -1, -1, -1), nb, tb), tb, cc


def operating_point(sd, hw=8):
    """{site: (median |mean|/std over groups or tokens, max of it, max |x|)} of the fp32 oracle at hw x hw."""
    stats = {}

    def probe(kind, name, x):
        if kind == "gn":
            g = x.reshape(x.shape[0], 32, -1)
        else:
            g = x.reshape(-1, x.shape[-1])
        r = (g.mean(-1).abs() / g.std(-1)).flatten()
        stats[name] = (r.median().item(), r.max().item(), x.abs().max().item())
    noisy, tb, cc = _calib_inputs(hw)
    R.PROBE = probe
    try:
        with torch.no_grad():
            out = R.unet_forward(sd, noisy, tb, cc, autocast=False)
    finally:
        R.PROBE = None
    stats["__output__"] = (0.0, 0.0, out.abs().max().item())
    return stats


def build(seed: int = 0, sweeps: int = SWEEPS, verbose: bool = False):
    """name -> fp32 torch tensor holding fp16-representable values (like `sd15_weights_torch`)."""
    sd = {k: torch.from_numpy(v).float() for k, v in synth.synth_state_dict(seed=seed, dtype=np.float16, mode="stress", stress_bias=0.0).items()}
    stream = _stream_writers()
    base = {}                                        # bias name -> its zero-offset value
    offsets = {}                                     # bias name -> offset vector currently inside sd[bias name] (= sd - base)
    noisy, tb, cc = _calib_inputs(8)
    for sweep in range(sweeps):
        order, sigma = [], {}

        def probe(kind, name, x):
            w = site_writer(name, stream)
            off = offsets.get(w)
            if kind == "gn":
                if off is not None and off.numel() == x.shape[1]:
                    x = x - off[None, :, None, None]
                elif off is not None:                # an up-block concat: the writer's channels are the leading ones
                    x = x.clone()
                    x[:, : off.numel()] -= off[None, :, None, None]
                s = x.reshape(x.shape[0], 32, -1).std(-1)
            else:
                if off is not None:
                    x = x - off
                s = x.reshape(-1, x.shape[-1]).std(-1)
            if name not in sigma:
                order.append(name)
            sigma[name] = s.median().item()
        R.PROBE = probe
        try:
            with torch.no_grad():
                R.unet_forward(sd, noisy, tb, cc, autocast=False)
        finally:
            R.PROBE = None
        done = set()
        for site in order:
            w = site_writer(site, stream)
            if w in done:                            # one writer, two sites (a stream tensor read by a resnet and kept as a skip): first wins
                continue
            done.add(w)
            n = sd[w].numel()
            u = synth.hash_uniform(site + "#ratio", 2, seed)
            r = R_LO * (R_HI / R_LO) ** u[0]
            resnet_inner = site.endswith("norm2") and "transformer_blocks" not in site
            sign = 1.0 if (u[1] < 0.5 or not resnet_inner) else -1.0                # stream and token offsets share a sign: they add up
            new = torch.from_numpy(sign * r * sigma[site] * (1.0 + 0.05 * synth.hash_normal(w + "#b", n, seed))).float()
            base.setdefault(w, sd[w].clone())
            sd[w] = (base[w] + new).half().float()
            offsets[w] = sd[w] - base[w]
        if verbose:
            st = operating_point(sd, 8)
            med = np.array([v[0] for k, v in st.items() if k != "__output__"])
            print(f"sweep {sweep}: median |mean|/std over sites: q10 {np.quantile(med, .1):.1f} q50 {np.median(med):.1f} q90 {np.quantile(med, .9):.1f}; "
                  f"max |act| {max(v[2] for v in st.values()):.0f}")
    return sd


_CACHE = {}


def stress_weights_torch(seed: int = 0):
    if seed not in _CACHE:
        _CACHE[seed] = build(seed)
    return _CACHE[seed]


if __name__ == "__main__":
    import time
    t0 = time.time()
    sd = build(verbose=True)
    print(f"built in {time.time() - t0:.0f} s")
    for hw in (8, 16):
        st = operating_point(sd, hw)
        for kind, sel in (("resnet norm1", lambda k: k.endswith("norm1") and "resnets" in k), ("resnet norm2", lambda k: k.endswith("norm2") and "resnets" in k),
                          ("transformer norm (GN)", lambda k: k.endswith(".norm")), ("LayerNorm 1/2/3", lambda k: "transformer_blocks" in k)):
            med = np.array([v[0] for k, v in st.items() if sel(k)])
            mx = np.array([v[1] for k, v in st.items() if sel(k)])
            print(f"{hw}x{hw} {kind:22s} n={len(med):2d} median-over-groups |mean|/std: min {med.min():5.2f} q25 {np.quantile(med, .25):5.2f} "
                  f"q50 {np.median(med):5.2f} q75 {np.quantile(med, .75):5.2f} max {med.max():5.2f}; max over groups {mx.max():6.1f}")
        print(f"{hw}x{hw} max |activation| at a norm input {max(v[2] for v in st.values()):.0f}, |eps_hat| max {st['__output__'][2]:.2f}")
