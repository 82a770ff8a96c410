#!/usr/bin/env python
"""Phase timing of the persistent 256x320 igemm tile (debug build with -DDM_IGEMM_TIMING, tools/build_timing.sh):
   python tools/igemm_timing.py M K N epi [conv_hw]     (dense shapes; epi 1 = GEGLU; LayerNorm-folded when epi = 1)
Prints shader cycles per tile of block 77, wave 0: k-step bodies, waits at the k-step tops, epilogue, tile switch."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DM_ENGINE_LIB", os.path.join(ROOT, "diff-mining_amd", "lib", "libdm_timing.so"))
os.environ["DM_IGEMM_BIG"] = "1"
import torch  # noqa: E402
from tests import gpu_util as U  # noqa: E402

lib = U.E.load_library()
M, K, N, epi = (int(a) for a in sys.argv[1:5])
conv_hw = int(sys.argv[5]) if len(sys.argv) > 5 else 0
extra = sys.argv[6] if len(sys.argv) > 6 else ""
d = "cuda"
if conv_hw:
    x = (torch.randn(M // (conv_hw * conv_hw), conv_hw, conv_hw, K, device=d) * 0.5).half()
    w = (torch.randn(N, 9 * K, device=d) * (9 * K) ** -0.5).half()
    mode = 1
else:
    x = (torch.randn(1, 1, M, K, device=d) * 0.5).half()
    w = (torch.randn(N, K, device=d) * K ** -0.5).half()
    mode = 0
b = torch.zeros(N, device=d).half()
res = torch.randn(x.shape[0], x.shape[1], x.shape[2], N, device=d).half() if extra == "res" else None
temb = torch.randn(x.shape[0], N, device=d).half() if extra == "temb" else None


def run():
    return U.op_igemm(x, w, b, epi=epi, mode=mode, res=res, temb=temb)


for _ in range(2):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
out = (C.c_longlong * 8)()
assert lib.dm_debug_pers_timing(out) == 0
body, wait, epil, sw, tiles = (out[i] for i in range(5))
tiles = max(tiles, 1)
nk = (9 if conv_hw else 1) * K // 64
print(f"M={M} K={K} N={N} epi={epi} {extra} conv_hw={conv_hw}: {e0.elapsed_time(e1):.3f} ms; block 77 did {tiles} tiles of {nk} k steps")
print(f"  per tile: k-step bodies {body / tiles:9.0f} ({body / tiles / nk:6.0f} per step)   waits at step tops {wait / tiles:8.0f} "
      f"({wait / tiles / nk:5.0f} per step)   epilogue {epil / tiles:8.0f}   tile switch {sw / tiles:7.0f}   "
      f"total {(body + wait + epil + sw) / tiles:9.0f} cycles")
