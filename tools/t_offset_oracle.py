#!/usr/bin/env python
"""Is the +1e-5-of-the-mean-loss offset of T(x|c) in fp16 (tools/t_deviation_fold.py) a property of fp16-autocast arithmetic
as such?  CPU only, no engine: the fp16-autocast ORACLE (the restatement of the reference's arithmetic, oracle/unet_ref.py) against the fp32
oracle on the same images and draws — N = 10 draws x 2 prompts at 32 x 32, the unrounded fp32 losses of `SD.compute_loss` (compute.py:95-102).
Prints the signed (T_autocast - T_fp32) / mean loss per image and the running mean +- standard error.  Test infrastructure.

    python tools/t_offset_oracle.py [n_images] [first_image] > profiles/r04_T_offset_autocast_oracle.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from diff_mining_amd import synth  # noqa: E402
from oracle import unet_ref as R  # noqa: E402


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    N, hw = 10, 32
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = {k: torch.from_numpy(v).float() for k, v in synth.synth_state_dict(seed=0, dtype=np.float16).items()}
    xs, _, _, c = synth.synth_inputs(first + n_img, 1, hw, hw, latent_dtype=np.float32)
    xs, c = torch.from_numpy(xs), torch.from_numpy(c).float()
    # D.noising's draws (compute.py:115-124,139-141): the same for every image
    g = torch.Generator().manual_seed(42)
    noises, ts = [], []
    for _ in range(N):
        noises.append(torch.randn(1, 4, hw, hw, generator=g, dtype=torch.float32))
        ts.append(torch.randint(100, 700, (1,), generator=g).long())
    noises, ts = torch.cat(noises), torch.cat(ts)
    nb, tb = torch.cat([noises] * 2), torch.cat([ts] * 2)
    cc = torch.cat([c[k:k + 1].expand(N, -1, -1) for k in range(2)])
    d = []
    for i in range(first, first + n_img):
        t0 = time.time()
        x = xs[i:i + 1]
        with torch.no_grad():
            l32 = R.compute_loss(sd, x, nb, tb, cc, autocast=False, latent_dtype=torch.float32).double()
            lac = R.compute_loss(sd, x, nb, tb, cc, autocast=True, latent_dtype=torch.float32).double()
        T32 = (l32[N:] - l32[:N]).mean().item()          # rows: cond 0 (c) first, then cond 1 (null): T = mean(L_null - L_c)
        Tac = (lac[N:] - lac[:N]).mean().item()
        ml = l32.mean().item()
        d.append((Tac - T32) / ml)
        a = np.array(d)
        se = a.std(ddof=1) / np.sqrt(len(a)) if len(a) > 1 else float("nan")
        print(f"image {i:2d}: T fp32 {T32:+.6e}  autocast {Tac:+.6e}  mean loss {ml:.4f}  (T_ac - T_32) / mean loss {d[-1]:+.2e}   "
              f"grid rel-L2 {((lac - l32).norm() / l32.norm()).item():.2e}   running mean {a.mean():+.2e} +- {se:.1e}   [{time.time() - t0:.0f} s]", flush=True)


if __name__ == "__main__":
    main()
