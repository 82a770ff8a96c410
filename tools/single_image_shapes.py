#!/usr/bin/env python
"""Per-shape launch table of ONE `TypicalityScorer.compute_losses` call on one 64x64 latent (the reference's own call shape: N draws x 2
prompts, compute.py:134-160):   DM_PROF_DUMP=/tmp/s.txt python tools/single_image_shapes.py [N]; python tools/prof_shapes.py /tmp/s.txt 3"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from diff_mining_amd import synth  # noqa: E402
from diff_mining_amd.engine import UNetEngine  # noqa: E402
from diff_mining_amd.typicality import TypicalityScorer  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
eng = UNetEngine(0)
eng.load_state_dict(synth.synth_state_dict(seed=0, dtype=np.float16))
x, _, _, c = synth.synth_inputs(1, 1, 64, 64, latent_dtype=np.float32)
x, c = torch.from_numpy(x).cuda(), torch.from_numpy(c).cuda()
sc = TypicalityScorer(eng, seed=42, N=N, t_min=0.1, t_max=0.7)
noises, ts = sc.draw(x.shape)
noises, ts = noises.cuda(), ts.cuda()
sc.compute_losses(x, c, noises=noises, timesteps=ts, to_host=False)
eng.prof_enable(True)
eng.prof_read()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(3):
    sc.compute_losses(x, c, noises=noises, timesteps=ts, to_host=False)
torch.cuda.synchronize()
print(f"N = {N}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per image", eng.prof_read())
