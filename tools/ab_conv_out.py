#!/usr/bin/env python
"""conv_out + eps-MSE at the bench batch (160 x 64 x 64 x 320): the per-pixel gather kernel (option conv_out_rows = 0) against the
rows-staged-in-LDS kernel (1), interleaved on one box; L2 / MALL flushed before every launch or not."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import gpu_util as U

lib = U.E.load_library()
d = U.dev()
B, H, W, C0 = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (160, 64, 64, 320)
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.randn(B, H, W, C0, device=d, generator=g)).half()
w = (torch.randn(4, 9 * C0, device=d, generator=g) * (9 * C0) ** -0.5).half()
b = torch.zeros(4, device=d).half()
eps = torch.randn(B, 4, H, W, device=d, generator=g)
loss = torch.empty(B, 4, H, W, device=d)
big = torch.empty(1 << 28, dtype=torch.float16, device=d)
for flush in (0, 1):
    for rep in range(2):
        for v in (0, 1):
            lib.dm_set_option(b"conv_out_rows", v)
            ts = []
            for i in range(12):
                if flush:
                    big.fill_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert lib.dm_op_conv_out(U.stream(), U.ptr(x), U.ptr(w), U.ptr(b), U.ptr(eps), B, H, W, C0, U.ptr(loss), None) == 0
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts = sorted(ts[2:])
            print(f"flush={flush} conv_out_rows={v}: median {ts[len(ts) // 2] * 1e3:7.1f} us  min {ts[0] * 1e3:7.1f} us")
lib.dm_set_option(b"conv_out_rows", 1)
