#!/usr/bin/env python
"""Phase timing of the pipelined attention kernel (debug build with -DDM_ATTN_TIMING):
   python tools/attn_timing.py   (expects diff-mining_amd/lib/libdm_timing.so)"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DM_ENGINE_LIB", os.path.join(ROOT, "diff-mining_amd", "lib", "libdm_timing.so"))
import torch  # noqa: E402
from tests import gpu_util as U  # noqa: E402

lib = U.E.load_library()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 40
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
B, heads = 160, 8
Cc = heads * D
qkv = torch.randn(B, T, 3 * Cc, device="cuda").half()
q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]
for _ in range(2):
    U.op_attention(q, k, v, heads)
print('hipOccupancyMaxActiveBlocksPerMultiprocessor:', lib.dm_debug_attn_occupancy())
span = (C.c_ulonglong * 2)()
dbg_span = lib.dm_debug_attn80_span if D == 80 else lib.dm_debug_attn_span
dbg_timing = lib.dm_debug_attn80_timing if D == 80 else lib.dm_debug_attn_timing
dbg_span(span, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
U.op_attention(q, k, v, heads)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
dbg_span(span, 0)
ticks_per_ns = (span[1] - span[0]) / (ms * 1e6)
print(f"kernel {ms:.3f} ms, {span[1] - span[0]} ticks first-start..last-end -> {ticks_per_ns:.3f} ticks/ns")
out = (C.c_longlong * 16)()
assert dbg_timing(out) == 0
names = ["vmcnt wait", "barrier", "dma issue (D40) / rescale check (D80)", "phase A (QK || exp)", "V reads+max+wait", "phase B (PV)", "epilogue-pre", "prologue"]
nt = T // 64
for w in range(2):
    print(f"wave {w}: per-tile cycles (shader clock ticks; {nt} tiles)")
    tot = 0
    for i, n in enumerate(names):
        v_ = out[w * 8 + i]
        per = v_ / nt if i < 6 else v_
        tot += v_
        print(f"   {n:24s} {per:10.1f}")
    print(f"   total {tot}")
    blocks = (T // 128) * heads * B
    print(f"   kernel {ms:.3f} ms; blocks {blocks}; resident blocks/CU ~ {blocks * tot / (ms * 1e6 * ticks_per_ns) / 256:.2f}")
