"""CPU tier: host-side logic of the path — image sharding, the single all-gather (gloo, world 2),
the reference-surface mirror's draw order and file naming, golden-vector self-consistency."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diff_mining_amd import typicality as T
from oracle import unet_ref as R

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_shard_indices_match_reference_striding():
    # compute.py:339  subs[i::sub_split]
    items = list(range(19))
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            idx = T.shard_indices(len(items), r, world)
            assert idx == items[r::world]
            seen += idx
        assert sorted(seen) == items


def _worker(rank, world, port, n_items, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = T.shard_indices(n_items, rank, world)
    local = torch.tensor([float(i) * 0.5 + 1.0 for i in idx])        # fake per-image T(x|c)
    res = T.gather_scores(local, n_items, rank, world)
    if rank == 0:
        torch.save(res, out)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 7])
def test_all_gather_of_scores_world2(tmp_path, n_items):
    out = str(tmp_path / "res.pt")
    port = 29500 + (os.getpid() % 2000) + n_items
    mp.spawn(_worker, args=(2, port, n_items, out), nprocs=2, join=True)
    res = torch.load(out)
    assert torch.equal(res, torch.tensor([float(i) * 0.5 + 1.0 for i in range(n_items)]))


def test_gather_world1_is_identity():
    s = torch.tensor([3.0, 1.0, 2.0])
    assert torch.equal(T.gather_scores(s, 3, 0, 1), s)


def test_get_path_matches_reference_naming():
    # compute.py:162-163
    assert T.TypicalityScorer.get_path("/out/1970", "/d/cars/1970__a.jpg") == "/out/1970/1970__a.npy"
    assert T.TypicalityScorer.get_path("/out", "x/y/z.png") == "/out/z.npy"


def test_draw_order_matches_oracle_without_engine():
    """D.noising order: randn_like then randint, N times after manual_seed (compute.py:139-141)."""
    sc = T.TypicalityScorer.__new__(T.TypicalityScorer)
    sc.seed, sc.N, sc.t_min, sc.t_max, sc.num_train_timesteps, sc.generator_device = 42, 4, 0.1, 0.7, 1000, "cpu"
    n, t = sc.draw((1, 4, 8, 8))
    n2, t2 = R.draw_noise_and_timesteps((1, 4, 8, 8), 4, 0.1, 0.7, seed=42)
    assert torch.equal(n, n2) and torch.equal(t, t2)
    g = np.load(os.path.join(GOLDEN, "grid_8x8.npz"))
    assert np.array_equal(g["noises"], n.numpy()) and np.array_equal(g["timesteps"], t.numpy())


def test_golden_vectors_reproduce_from_oracle(sd15_weights_torch):
    """The committed fixtures are exactly what the oracle produces today (guards silent drift)."""
    g = np.load(os.path.join(GOLDEN, "score_8x8.npz"))
    x, eps, t, c = (torch.from_numpy(g[k]) for k in ("x", "eps", "t", "c"))
    nb, tb = torch.cat([eps] * 2), torch.cat([t] * 2)
    cc = torch.cat([c[k:k + 1].expand(2, -1, -1) for k in range(2)])
    l32 = R.compute_loss(sd15_weights_torch, x, nb, tb, cc, autocast=False)
    rel = ((l32 - torch.from_numpy(g["loss_fp32"])).norm() / l32.norm()).item()
    assert rel < 1e-5, rel
    la = R.compute_loss(sd15_weights_torch, x, nb, tb, cc, autocast=True)
    rel = ((la - torch.from_numpy(g["loss_autocast"])).norm() / la.norm()).item()
    assert rel < 5e-3, rel          # fp16 emulation is BLAS-order sensitive; fp32 row is the tight pin
    grid = np.load(os.path.join(GOLDEN, "grid_8x8.npz"))["grid"]
    assert grid.shape == (4, 2, 4, 8, 8) and grid.dtype == np.float16
