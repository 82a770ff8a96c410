#!/usr/bin/env python
"""Phase timing of the 256x320 igemm tile kernel (debug build with -DDM_IGEMM_TIMING, tools/build_timing.sh):
   python tools/igemm_timing.py M K N epi     (dense shapes; epi 1 = GEGLU)"""
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
conv_hw = int(sys.argv[5]) if len(sys.argv) > 5 else 0        # conv3x3 on [M/hw^2, hw, hw, K] instead of a dense layer
if conv_hw:
    x = (torch.randn(M // (conv_hw * conv_hw), conv_hw, conv_hw, K, device="cuda") * 0.5).half()
    w = (torch.randn(N, 9 * K, device="cuda") * (9 * K) ** -0.5).half()
    mode = 1
else:
    x = (torch.randn(1, 1, M, K, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    mode = 0
b = torch.zeros(N, device="cuda").half()
for _ in range(2):
    U.op_igemm(x, w, b, epi=epi, mode=mode)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
U.op_igemm(x, w, b, epi=epi, mode=mode)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f"M={M} K={K} N={N} epi={epi}: {ms:.3f} ms, {2.0 * M * N * K * (9 if conv_hw else 1) / ms / 1e9:.0f} TF/s")
out = (C.c_longlong * 16)()
if not hasattr(lib, "dm_debug_igemm_timing"):
    raise SystemExit(0)
assert lib.dm_debug_igemm_timing(out) == 0
tiles_per_cu = ((M + 255) // 256) * (N // 320) / 256
print(f"M={M} K={K} N={N} epi={epi}: {ms:.3f} ms, {2.0 * M * N * K * (9 if conv_hw else 1) / ms / 1e9:.0f} TF/s, {tiles_per_cu:.1f} tiles/CU")
names = ["setup", "k waits", "k bodies", "epi rest", "epi: first barrier(s)", "epi: convert+ds_write", "epi: barrier 2", "epi: readback+store"]
for wv in range(2):
    vals = [out[wv * 8 + i] for i in range(8)]
    print(f"  wave {wv}: " + ", ".join(f"{n}={v}" for n, v in zip(names, vals)) + f", total={sum(vals)}")
