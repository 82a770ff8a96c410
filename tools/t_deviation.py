#!/usr/bin/env python
"""Where the engine's T(x|c) sits at the BASELINE draw count (N = 10 draws x 2 prompts, 32x32 latents), on the GPU.

For n images (default 6) of tests/test_gpu_e2e.py::test_score_at_baseline_draw_count's inputs:
  T of the fp32 oracle (ground truth), of the fp16-autocast oracle, and of the engine — default options, and with single options
  switched off (folds, tap reuse, the pipelined attention kernels) to see whether any rewrite moves T systematically.
Prints per image |dT|/|T| and |dT|/mean loss of (engine vs autocast oracle), (engine vs fp32 oracle), (autocast oracle vs fp32
oracle), next to the autocast oracle's own spread under re-ordering (tests/golden/oracle_T_floor.json).  Test infrastructure
(runs the oracle on the host cores).

    python tools/t_deviation.py [n_images] [--fp32-only] > profiles/r04_T_deviation.txt
(--fp32-only: the fp32 oracle alone, fewer option variants — more images per minute; the "vs autocast" columns then repeat "vs fp32")
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from diff_mining_amd import synth  # noqa: E402
from diff_mining_amd.engine import UNetEngine  # noqa: E402
from diff_mining_amd.typicality import TypicalityScorer  # noqa: E402
from oracle import unet_ref as R  # noqa: E402

OPTIONS = [(), ("ln_fold",), ("gn_fold",), ("ff_fold",), ("sc_fold",), ("tap_reuse",), ("attn_pipe",), ("attn_cross",), ("ln_inkernel",),
           ("ln_fold", "gn_fold", "ff_fold", "sc_fold", "tap_reuse")]
DEFAULTS = {"ln_fold": 1, "gn_fold": 1, "ff_fold": 1, "sc_fold": 1, "tap_reuse": 1, "attn_pipe": 1, "attn_cross": 1, "ln_inkernel": 1}


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    fp32_only = "--fp32-only" in sys.argv           # more images, ground truth only (the autocast emulation is the slow oracle)
    global OPTIONS
    if fp32_only:
        OPTIONS = [(), ("sc_fold",), ("ln_inkernel",), ("ln_fold", "gn_fold", "ff_fold", "sc_fold", "tap_reuse")]
    N, hw = 10, 32
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sdn = synth.synth_state_dict(seed=0, dtype=np.float16)
    sd = {k: torch.from_numpy(v).float() for k, v in sdn.items()}
    eng = UNetEngine(0)
    eng.load_state_dict(sdn)
    sc = TypicalityScorer(eng, seed=42, N=N, t_min=0.1, t_max=0.7)
    xs, _, _, c = synth.synth_inputs(n_img, 1, hw, hw, latent_dtype=np.float32)
    xs, c = torch.from_numpy(xs), torch.from_numpy(c)
    floor = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_T_floor.json")))
    print(f"autocast oracle vs itself under re-ordering (tests/golden/oracle_T_floor.json, 2 images x 3 re-orderings): |dT|/|T| "
          f"{min(v['dT_over_T'] for im in floor['images'] for v in im['variants'].values()):.2e} .. {floor['max_dT_over_T']:.2e}, "
          f"|dT|/mean loss .. {floor['max_dT_over_mean_loss']:.2e}")
    rows = {o: [] for o in OPTIONS}
    ac32 = []
    for i in range(n_img):
        x = xs[i:i + 1]
        noises, ts = sc.draw(x.shape)
        with torch.no_grad():
            g_32 = R.compute_losses(sd, x, c.float(), noises, ts, B=N, autocast=False)
            g_ac = g_32 if fp32_only else R.compute_losses(sd, x, c.float(), noises, ts, B=N, autocast=True, latent_dtype=torch.float32)
        T_ac, T_32 = R.typicality_scalar(g_ac).item(), R.typicality_scalar(g_32).item()
        ml = g_32.float().mean().item()
        ac32.append((abs(T_ac - T_32) / abs(T_32), abs(T_ac - T_32) / ml))
        print(f"image {i}: T fp32 oracle {T_32:+.6e}  autocast oracle {T_ac:+.6e}  mean loss {ml:.4f}   autocast vs fp32: |dT|/|T| {ac32[-1][0]:.2e}  "
              f"|dT|/mean loss {ac32[-1][1]:.2e}")
        for off in OPTIONS:
            try:
                for o in off:
                    assert eng.lib.dm_set_option(o.encode(), 0) == 0
                grid = sc.compute_losses(x, c, noises=noises, timesteps=ts, to_host=False)
                T = eng.reduce_typicality(grid)[1].item()
            finally:
                for o in off:
                    eng.lib.dm_set_option(o.encode(), DEFAULTS[o])
            rl = ((grid.float().cpu() - g_ac.float()).norm() / g_ac.float().norm()).item()
            rows[off].append((abs(T - T_ac) / abs(T_ac), abs(T - T_ac) / ml, abs(T - T_32) / abs(T_32), abs(T - T_32) / ml, rl, (T - T_32) / abs(T_32)))
            print(f"image {i}:   engine [{'default' if not off else ' '.join(off) + ' off':44s}] T {T:+.6e}  vs autocast |dT|/|T| {rows[off][-1][0]:.2e} "
                  f"|dT|/ml {rows[off][-1][1]:.2e}   vs fp32 |dT|/|T| {rows[off][-1][2]:.2e} |dT|/ml {rows[off][-1][3]:.2e}   grid rel-L2 {rl:.2e}")
    a = np.array(ac32)
    print(f"\nsummary over {n_img} images (rms / max)")
    print(f"  autocast oracle vs fp32 oracle          : |dT|/|T| {np.sqrt((a[:, 0] ** 2).mean()):.2e} / {a[:, 0].max():.2e}   |dT|/mean loss "
          f"{np.sqrt((a[:, 1] ** 2).mean()):.2e} / {a[:, 1].max():.2e}")
    for off in OPTIONS:
        r = np.array(rows[off])
        print(f"  engine [{'default' if not off else ' '.join(off) + ' off':44s}] vs autocast: |dT|/|T| {np.sqrt((r[:, 0] ** 2).mean()):.2e} / {r[:, 0].max():.2e}  "
              f"|dT|/ml {np.sqrt((r[:, 1] ** 2).mean()):.2e} / {r[:, 1].max():.2e}   vs fp32: |dT|/|T| {np.sqrt((r[:, 2] ** 2).mean()):.2e} / {r[:, 2].max():.2e}  "
              f"|dT|/ml {np.sqrt((r[:, 3] ** 2).mean()):.2e} / {r[:, 3].max():.2e}   signed mean (T - T32)/|T32| {r[:, 5].mean():+.2e}")


if __name__ == "__main__":
    main()
