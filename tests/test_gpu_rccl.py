"""First contact of the multi-GPU leg with RCCL on a 1-GPU box (`init_process_group("nccl")` and the
all-gather had only ever run on gloo).  SURVEY 8e / compute.py:337-341: image shards `r::world`, ONE all-gather of T(x|c).

A world of one rank is all a 1-GPU box offers: it still loads librccl, creates the communicator on 127.0.0.1 with the
dmabuf IPC mode the pool needs, and runs the very calls of the N > 1 path — `_all_gather_padded` (the collective behind
`gather_scores` / `gather_grids`), `dist.barrier`, the float64 MAX all-reduce and per-rank all-gather of bench.py's timing.
Each case runs in a subprocess so no process group leaks into the other GPU tests."""
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, extra_env=None, timeout=600):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29541", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", "")})
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    # librccl writes a banner ("Librccl path : ...") to the C-level stdout, flushed at exit, i.e. after the JSON line
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_rccl_all_gather_of_scores_world1():
    assert torch.cuda.is_available()
    out = _run("""
        import json, torch, torch.distributed as dist
        from diff_mining_amd.typicality import _all_gather_padded
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        dev = torch.device("cuda", 0)
        s = torch.arange(8, dtype=torch.float32, device=dev) * 0.5 - 1.0         # 8 per-image T(x|c) scalars
        g = _all_gather_padded(s, 8, 1)                                          # the collective of gather_scores
        grids = (torch.arange(3 * 2 * 2 * 4 * 8 * 8, device=dev) % 2048).to(torch.float16).view(3, 2, 2, 4, 8, 8)
        gg = _all_gather_padded(grids, 4, 1)                                     # gather_grids: ragged shard padded to 4 rows
        dist.barrier()
        tt = torch.tensor([1.25], dtype=torch.float64, device=dev)               # bench.py's timing collectives
        each = [torch.zeros_like(tt)]
        dist.all_gather(each, tt)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        ok = bool(torch.equal(g[0], s)) and bool(torch.equal(gg[0, :3], grids)) and float(gg[0, 3].abs().sum()) == 0.0
        print(json.dumps({"ok": ok, "backend": dist.get_backend(), "world": dist.get_world_size(),
                          "each": each[0].item(), "max": tt.item(),
                          "rccl": ".".join(str(v) for v in torch.cuda.nccl.version())}))
        dist.destroy_process_group()
    """)
    assert out["ok"] and out["backend"] == "nccl" and out["world"] == 1
    assert out["each"] == 1.25 and out["max"] == 1.25
    print("RCCL", out["rccl"])


def test_bench_two_ranks_with_real_engines_on_one_gpu():
    """Rehearsal of the driver's `bench.py --gpus N` on the 1-GPU box (r06): `DM_BENCH_ONE_GPU=1` runs N = 2 ranks with REAL engines, both on
    cuda:0, over gloo (RCCL refuses two ranks on one device; its own calls are the test above).  Everything else is the N > 1 path as
    the driver will run it: bench.py launching its own ranks, rank 0 writing the 1.7 GB weight slab to /dev/shm and rank 1 mapping it, two
    engines, the r::2 image shards, the all-gather of T(x|c) in global image order, the barrier / MAX timing, ONE JSON line — and the slab gone."""
    assert torch.cuda.is_available()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"DM_BENCH_ONE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    before = set(os.listdir("/dev/shm")) if os.path.isdir("/dev/shm") else set()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-side",
                        "--no-parity"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["backend"] == "gloo" and "rehearsal" in out
    assert len(out["rank_ms_per_step"]["all"]) == 2 and out["ms_per_step"] >= out["rank_ms_per_step"]["min"] - 1e-3
    assert out["value"] > 0 and out["scores_checksum"] == out["scores_checksum"] and abs(out["scores_checksum"]) < 10.0      # 16 finite T(x|c)
    assert out["config"]["parallelism"].startswith("image-sharded x2")
    after = set(os.listdir("/dev/shm")) if os.path.isdir("/dev/shm") else set()
    assert not [f for f in after - before if f.startswith("dm_bench_weights_")], after - before
