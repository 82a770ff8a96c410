#!/usr/bin/env python
"""Per-kernel time of the single-pass GroupNorm against the two-pass pair on the bench's shapes (run under rocprofv3 --kernel-trace):
    rocprofv3 --kernel-trace -f csv -d /tmp/gn -o t -- python tools/gn_fused_time.py ; python tools/gn_fused_time.py parse /tmp/gn"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(160, 64, 64, 320, 0), (160, 64, 64, 320, 320), (160, 32, 32, 640, 0), (160, 32, 32, 640, 640), (160, 32, 32, 1280, 640),
          (160, 16, 16, 1280, 0), (160, 16, 16, 1280, 1280), (160, 8, 8, 1280, 1280)]

if len(sys.argv) > 1 and sys.argv[1] == "parse":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "gn_" in r["Kernel_Name"]]
    t = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    i = 0
    for (N, H, W, C1, C2) in SHAPES:
        st, ap = t(rows[i]), t(rows[i + 1])
        fu = [t(r) for r in rows[i + 2:i + 5]]
        i += 5
        gb = N * H * W * (C1 + C2) * 2 / 1e9
        print(f"N={N} {H}x{W} C={C1}+{C2}: two-pass {st:.1f} + {ap:.1f} = {st + ap:.1f} us ({3 * gb / (st + ap) * 1e3:.0f} GB/s of 3 passes); "
              f"single pass {min(fu):.1f} us ({2 * gb / min(fu) * 1e3:.0f} GB/s of 2 passes)  ratio {min(fu) / (st + ap):.2f}")
    sys.exit(0)

import torch  # noqa: E402
from diff_mining_amd import engine as E  # noqa: E402
from tests import gpu_util as U  # noqa: E402

lib = E.load_library()
d = torch.device("cuda", 0)
for (N, H, W, C1, C2) in SHAPES:
    x1 = torch.randn(N, H, W, C1, device=d).half()
    x2 = torch.randn(N, H, W, C2, device=d).half() if C2 else None
    Ct = C1 + C2
    g, b = torch.ones(Ct, device=d), torch.zeros(Ct, device=d)
    y = torch.empty(N, H, W, Ct, dtype=torch.float16, device=d)
    assert lib.dm_op_groupnorm(U.stream(), U.ptr(x1), U.ptr(x2), N, H * W, Ct, C1, 32, 1e-5, U.ptr(g), U.ptr(b), 1, U.ptr(y)) == 0
    for _ in range(3):
        rc = lib.dm_op_groupnorm_fused(U.stream(), U.ptr(x1), U.ptr(x2), N, H * W, Ct, C1, 32, 1e-5, U.ptr(g), U.ptr(b), 1, U.ptr(y))
        assert rc in (0, 4), rc
    torch.cuda.synchronize()
