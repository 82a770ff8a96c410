#!/usr/bin/env python
"""Golden vectors for the CONSUMERS of the loss grids, produced by the reference's own code run in this container.

The arithmetic of the U-Net lives in diffusers (absent here), but the reductions that consume `[N, 2, 4, h, w]` grids are
plain torch / numpy code inside the reference repository:

    normalize                    diffmining/typicality/cluster.py:32-47 and diffmining/typicality/utils.py:14-20
    pool                         diffmining/typicality/utils.py:74-80
    d_compute                    diffmining/typicality/utils.py:122-134
    Cluster.load_typicality_norm diffmining/typicality/cluster.py:112-123
    Cluster.load_typicality      diffmining/typicality/cluster.py:125-137
    Cluster.rank_images.compute  diffmining/typicality/cluster.py:517-531
    <X-ray class>.compute        diffmining/applications/xray/compute.py:210-218 (dm_pixel)

The two modules cannot be imported as they are (`skimage`, `umap` are not installed), so this script parses them with `ast`,
compiles exactly those function definitions — the reference's text, unmodified, never written anywhere — and calls them with
seeded synthetic grids; the methods get a duck-typed `self` (device, kx, ky, an image size) and `d(path)` returning the grid.
Only inputs and outputs are stored: tests/golden/consumers_ref.npz.  Needs /root/reference (this container only); the tests
read the .npz.

    python tests/make_golden_consumers.py
"""
import ast
import os
import sys
import types

import numpy as np
import torch
from torch.nn.functional import interpolate

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def ref_function(relpath, path, ns):
    src = open(os.path.join(REF, relpath)).read()
    tree = ast.parse(src)
    node = tree
    for name in path:
        node = next(ch for ch in ast.iter_child_nodes(node)
                    if isinstance(ch, (ast.FunctionDef, ast.ClassDef)) and ch.name == name)
    mod = ast.Module(body=[node], type_ignores=[])
    exec(compile(mod, os.path.join(REF, relpath), "exec"), ns)
    return ns[path[-1]]


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    CL, UT = "diffmining/typicality/cluster.py", "diffmining/typicality/utils.py"
    base = {"np": np, "torch": torch, "interpolate": interpolate}
    u_ns = dict(base)
    pool = ref_function(UT, ("pool",), u_ns)
    u_normalize = ref_function(UT, ("normalize",), u_ns)
    d_compute = ref_function(UT, ("d_compute",), u_ns)
    c_ns = dict(base, pool=pool)
    c_normalize = ref_function(CL, ("normalize",), c_ns)
    load_typicality_norm = ref_function(CL, ("Cluster", "load_typicality_norm"), c_ns)
    load_typicality = ref_function(CL, ("Cluster", "load_typicality"), c_ns)

    XR = "diffmining/applications/xray/compute.py"
    x_ns = dict(base)
    xray_compute = None
    for cls in [n for n in ast.parse(open(os.path.join(REF, XR)).read()).body if isinstance(n, ast.ClassDef)]:
        if any(isinstance(n, ast.FunctionDef) and n.name == "compute" and [a.arg for a in n.args.args][:3] == ["self", "dm", "size"] for n in cls.body):
            xray_compute = ref_function(XR, (cls.name, "compute"), x_ns)        # `compute(self, dm, size, blur=False)` (xray/compute.py:210-218)
    assert xray_compute is not None
    out = {}
    rng = np.random.default_rng(20260929)
    cases = {"a": (3, 12, 10, 45, 37, 5), "b": (10, 16, 21, 128, 171, 32), "c": (2, 16, 16, 96, 96, 1)}
    for tag, (N, h, w, H, W, k) in cases.items():
        # losses are positive, cond and null close to each other (as real grids): 1 + noise, fp16 like D.compute_losses
        grid = (1.0 + 0.3 * rng.standard_normal((N, 2, 4, h, w)) + 0.05 * rng.standard_normal((1, 2, 1, h, w))).astype(np.float16)
        out[f"{tag}_grid"] = grid
        out[f"{tag}_size"] = np.array([H, W, k], dtype=np.int64)

        class _Img:
            size = (W, H)                                   # PIL: (width, height)
        me = types.SimpleNamespace(device="cpu", kx=k, ky=k, load_image=lambda path: _Img)
        d = lambda path: grid.copy()                          # noqa: E731  (`d(path)` = np.load of the stored grid)
        out[f"{tag}_load_typicality"] = np.asarray(load_typicality(me, d, "x.jpg"), dtype=np.float32)
        me1 = types.SimpleNamespace(device="cpu", kx=1, ky=1, load_image=lambda path: _Img)
        out[f"{tag}_load_typicality_k1"] = np.asarray(load_typicality(me1, d, "x.jpg"), dtype=np.float32)
        dmn = load_typicality_norm(me, d, "x.jpg")
        assert dmn.dtype == np.float32
        out[f"{tag}_load_typicality_norm"] = dmn
        box = (H // 4, W // 5, H // 4 + max(2, H // 3), W // 5 + max(2, W // 2))
        out[f"{tag}_box"] = np.array(box, dtype=np.int64)
        dc = d_compute(grid.copy(), H, W, *box)
        assert dc.dtype == np.float32
        out[f"{tag}_d_compute"] = dc
        # rank_images' per-image scalar (cluster.py:517-531): the nested function, with its closure variables supplied
        r_ns = dict(base, self=me, d=lambda p: grid.copy())
        rank = ref_function(CL, ("Cluster", "rank_images", "compute"), r_ns)
        _, score = rank(("x.jpg", True))
        out[f"{tag}_rank_score"] = np.array(score, dtype=np.float32)
        # the X-ray application's per-pixel map (applications/xray/compute.py:210-218): `dm_pixel`, second return value
        _, dm_pixel = xray_compute(types.SimpleNamespace(), torch.from_numpy(grid.copy()), (H, W))
        out[f"{tag}_xray_dm_pixel"] = np.asarray(dm_pixel, dtype=np.float32)
        if tag != "a":
            continue
        dm = out[f"{tag}_load_typicality_k1"]
        out[f"{tag}_cnorm_positive"] = c_normalize(dm.copy(), positive_only=True)
        sp = c_normalize(dm.copy(), positive_only="split")
        out[f"{tag}_cnorm_split_pos"], out[f"{tag}_cnorm_split_neg"] = sp
        out[f"{tag}_unorm"] = u_normalize(dm.copy())
        out[f"{tag}_unorm_positive"] = u_normalize(dm.copy(), positive_only=True)
    p = os.path.join(HERE, "golden", "consumers_ref.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, {k: v.shape for k, v in out.items() if k.startswith("a_")}, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
