import sys, torch
sys.path.insert(0, '.')
from tests import gpu_util as U
lib = U.E.load_library()
d = U.dev()
H, D, T = 8, 160, 256
for B in (8, 32, 64, 160, 320):
    C = H * D
    qkv = torch.randn(B, T, 3 * C, device=d).half()
    q, k, v = qkv[..., :C], qkv[..., C:2*C], qkv[..., 2*C:]
    for opt in (9, 1):
        lib.dm_set_option(b"attn_pipe", opt)
        for _ in range(3): U.op_attention(q, k, v, H)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): U.op_attention(q, k, v, H)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        fl = 4.0 * B * H * T * T * D
        by = 4.0 * B * T * C * 2
        print(f"B={B:4d} attn_pipe={opt}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s  {by/ms/1e9:6.2f} TB/s (Q+K+V+O)")
lib.dm_set_option(b"attn_pipe", 1)
