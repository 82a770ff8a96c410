#!/usr/bin/env python
"""bench.py — typicality-scored images/sec/node (BASELINE.json metric) on N MI355X of one node.

One "step" = one pass of the hot path over one batch of synthetic input PER GPU:
    8 images of 512x512 (latent 4x64x64), each under ITS OWN category prompt and the shared null prompt (the reference's work
    list: one `path,category` line per image, compute.py:284-290), each scored with 10 (t, eps) draws x 2 prompts
    = 160 SDv1.5 U-Net forwards of the fused add_noise -> U-Net -> eps-MSE path, through the PRODUCT SURFACE
    `TypicalityScorer.compute_losses_batch` -> [8, 10, 2, 4, 64, 64] fp16 grids (dm_score_conds_slots: the reference's
    draw-tiled-over-prompts batch with a per-image prompt-slot table, prompt-independent head computed once per draw), then ONE
    on-device typicality reduction over all images (dm_reduce_typicality_batched); N > 1: plus ONE all-gather
    of the per-image T(x|c) scalars (RCCL over xGMI).  This is BASELINE.json configs[1] (and [2] for N>1).
Inputs (fp32 latents and draws like the reference's, fp16 prompt embeddings) and the synthetic fp16 weights are
resident in HBM before the timed region.  Weak scaling: every rank scores its own 8 images.

`python bench.py --gpus N` launches its own N ranks (re-exec under torch.distributed.run on 127.0.0.1) when it is
not already inside a launcher; under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
it uses the ranks it was given and refuses a WORLD_SIZE that contradicts --gpus.  One process per GPU, image
sharding `r::N` as the reference's `subs[i::sub_split]` (diffmining/typicality/compute.py:337-341).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     — dominant kernel = the implicit-GEMM MFMA kernel family (84 % of the FLOPs; igemm_pers_kernel + igemm_kernel):
                 achieved = the multiply-adds its launches ISSUED / their summed kernel time, measured with HIP events
                 on the launch stream over the timed steps (dm_prof_*); peak = 2.5 PFLOP/s dense fp16.  `achieved_nominal` books
                 SURVEY 8d's algorithmic count instead (Upsample2D.conv = interpolate + nine taps; `up_fold` issues 4/9 of that).
  single_image_call — images/s of `TypicalityScorer.compute_losses` on ONE image (the reference's own call, compute.py:134-160) at
                 N = 10 draws (20 U-Net samples per engine call) and at the reference default N = 100 (compute.py:106).
  side_workloads — BASELINE configs[3] (DIFT-161 in the reference's fp32) and configs[4] (X-ray 1024 px heat-map), 3 steps each.
  cpu_baseline — the oracle (fp32 PyTorch-CPU restatement; kind "port") timed on this host's cores on
                 one full image of the workload (20 U-Net forwards @64x64, ~45 s).
`--workload dift|xray` print the same keys for BASELINE configs[3] / [4].
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOP_PER_FORWARD_64 = 803.27e9          # SURVEY.md §8d (2 FLOP/MAC, attention included), nominal
N_IMG, N_DRAWS, N_COND, LAT = 8, 10, 2, 64
PEAK_TFLOPS = 2500.0                    # MI355X dense fp16 MFMA (MI355X_MICROARCH.md)
SCORE_DEVIATION_MAX = 1.0e-3            # north_star: "<= 1e-3 score deviation from reference" on the loss grid (vs exact fp32, same inputs)
PEAK_TFLOPS_F32 = 157.3                 # MI355X fp32 matrix (v_mfma_f32_*_f32: 256 FLOP/clk/CU; MI355X_MICROARCH.md "Peak FP32 (matrix)")
STUB = os.environ.get("DM_BENCH_STUB", "0") not in ("", "0")     # CPU test of the launcher / gather path (gloo, no engine)
# Rehearsal of the N > 1 path on a ONE-GPU box (r06): N ranks with REAL engines, all on cuda:0, gloo instead of RCCL (RCCL refuses two ranks on
# one device).  Everything the driver's `--gpus N` run goes through except the RCCL transport itself (tests/test_gpu_rccl.py covers that at
# world 1): the launcher, the /dev/shm weight slab, N engines, the r::N image shards, the gather, the barrier / MAX timing, the JSON line.
# The line says so (`rehearsal`); its rate is NOT a scaling number (the ranks share one GPU).
ONE_GPU = os.environ.get("DM_BENCH_ONE_GPU", "0") not in ("", "0")


class ClockSampler:
    """Shader clock and socket power of this rank's GPU, sampled in a thread while the timed steps run (the chip sits
    at its 1.4 kW cap and clocks ~2.0 GHz under this load, so a fraction of the NOMINAL 2.4 GHz peak mixes kernel inefficiency with
    clock).  Sources, first that answers: amdsmi's gpu_metrics (current_gfxclk(s), current / average socket power), amdsmi's
    clock_info(SYS) + power_info, the hwmon files under /sys/class/drm.  A box where none answers yields nulls and says why."""

    def __init__(self, index: int = 0, period_s: float = 0.05):
        import threading
        self.index, self.period, self.samples, self.source, self.error = index, period_s, [], None, None
        self._stop, self._thread, self._read = threading.Event(), None, None
        self._pick_source()

    @staticmethod
    def _num(v):
        try:
            v = float(v)
        except (TypeError, ValueError):
            return None
        return v if 0.0 < v < 60000.0 else None            # "N/A", 0 and the 0xFFFF placeholder are not readings

    def _pick_source(self):
        errs = []
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            h = hs[self.index if self.index < len(hs) else 0]

            def metrics():
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                cl = [self._num(c) for c in (m.get("current_gfxclks") or [])]
                cl = [c for c in cl if c is not None]
                clk = sum(cl) / len(cl) if cl else self._num(m.get("current_gfxclk")) or self._num(m.get("average_gfxclk_frequency"))
                pw = self._num(m.get("current_socket_power")) or self._num(m.get("average_socket_power"))
                return clk, pw

            def clock_power():
                c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.SYS)
                w = amdsmi.amdsmi_get_power_info(h)
                return (self._num(c.get("clk")) or self._num(c.get("cur_clk")),
                        self._num(w.get("current_socket_power")) or self._num(w.get("average_socket_power")))
            for name, fn in (("amdsmi.gpu_metrics", metrics), ("amdsmi.clock_info+power_info", clock_power)):
                try:
                    if fn()[0] is not None:
                        self._read, self.source = fn, name
                        return
                    errs.append(f"{name}: no clock value")
                except Exception as ex:                      # noqa: BLE001  (a missing reading must not cost the measurement)
                    errs.append(f"{name}: {type(ex).__name__}")
        except Exception as ex:                              # noqa: BLE001
            errs.append(f"amdsmi: {type(ex).__name__}")
        try:
            import glob
            cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
            hw = cards[self.index if self.index < len(cards) else 0]

            def hwmon():
                def rd(n):
                    try:
                        with open(os.path.join(hw, n)) as f:
                            return float(f.read().strip())
                    except Exception:                        # noqa: BLE001
                        return None
                f1, pw = rd("freq1_input"), rd("power1_input") or rd("power1_average")
                return (self._num(f1 / 1e6) if f1 else None, self._num(pw / 1e6) if pw else None)
            if hwmon()[0] is not None:
                self._read, self.source = hwmon, "sysfs hwmon"
                return
            errs.append("hwmon: no freq1_input")
        except Exception as ex:                              # noqa: BLE001
            errs.append(f"hwmon: {type(ex).__name__}")
        try:
            import re
            import shutil
            exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"

            def cli():                                         # slow (a process per sample): the last resort
                txt = subprocess.run([exe, "-d", str(self.index), "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
                c = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", txt)
                w = re.search(r"Socket Graphics Package Power \(W\):\s*([0-9.]+)", txt)
                return (self._num(c.group(1)) if c else None, self._num(w.group(1)) if w else None)
            if cli()[0] is not None:
                self._read, self.source = cli, "rocm-smi --showclocks --showpower (one process per sample)"
                return
            errs.append("rocm-smi: no sclk line")
        except Exception as ex:                              # noqa: BLE001
            errs.append(f"rocm-smi: {type(ex).__name__}")
        self.error = "; ".join(errs)

    def _run(self):
        while not self._stop.is_set():
            try:
                c, w = self._read()
                if c is not None:
                    self.samples.append((c, w))
            except Exception:                                # noqa: BLE001
                pass
            self._stop.wait(self.period)

    def start(self):
        if self._read is not None:
            import threading
            self.samples = []
            self._stop.clear()
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self) -> dict:
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None
        cl = [c for c, _ in self.samples]
        pw = [w for _, w in self.samples if w is not None]
        if not cl:
            return {"sclk_mhz_mean": None, "power_w_mean": None, "clock_samples": 0, "clock_source": self.source,
                    "clock_note": self.error or "no sample fell inside the timed region"}
        return {"sclk_mhz_mean": round(sum(cl) / len(cl), 1), "sclk_mhz_min": round(min(cl), 1), "sclk_mhz_max": round(max(cl), 1),
                "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "power_w_max": round(max(pw), 1) if pw else None,
                "clock_samples": len(cl), "clock_source": self.source}


NOMINAL_SCLK_MHZ = 2400.0               # the clock PEAK_TFLOPS is quoted at (MI355X_MICROARCH.md: max clock 2400 MHz)


def clock_normalised(smi: dict, achieved_tflops: float, peak_tflops: float) -> dict:
    """`frac_at_sustained_clock` = achieved / (peak x mean sclk / 2400 MHz): the share of what the matrix cores can do at the clock the
    chip actually held over the timed steps (beside `frac`, which prices against the nominal 2.4 GHz peak)."""
    out = dict(smi)
    clk = smi.get("sclk_mhz_mean")
    out["peak_at_sustained_clock"] = round(peak_tflops * clk / NOMINAL_SCLK_MHZ, 1) if clk else None
    out["frac_at_sustained_clock"] = round(achieved_tflops / (peak_tflops * clk / NOMINAL_SCLK_MHZ), 4) if clk else None
    return out


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args) -> int:
    """`bench.py --gpus N` outside a launcher: start N ranks of this script, one per GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def SLAB_PATHS() -> list:
    # one name per launch (the launcher's port is unique per run on this node): two benches on one node do not collide.
    # /dev/shm first (pages shared between the ranks, no disk), /tmp if the node's shm mount is too small or absent
    name = f"dm_bench_weights_{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}.slab"
    return [os.path.join(d, name) for d in ("/dev/shm", "/tmp") if os.path.isdir(d)]


def SLAB_PATH() -> str:
    return SLAB_PATHS()[0]


_slab_in_use = None        # the path rank 0 wrote (every rank knows it after node_state_dict), None = no slab this run


def node_state_dict(rank: int, world: int, make=None):
    """The synthetic fp16 U-Net weights of this run.  One rank: synthesised in place.  N ranks of a node: rank 0 synthesises once and
    writes ONE 1.7 GB slab to /dev/shm, the others map it (same bytes by construction — the generator is a pure function of
    (seed, name, index) — and 1/N of the host arithmetic and memory); removed again once every rank has loaded (main).
    A node whose /dev/shm and /tmp both refuse the slab (a 64 MB container shm, a full disk) is not an error: rank 0 tells the
    others through the process group and every rank synthesises its own copy — the same bytes, N times the host work."""
    global _slab_in_use
    from diff_mining_amd import synth
    import torch.distributed as dist
    make = make or (lambda: synth.synth_state_dict(seed=0, dtype=np.float16))
    if world == 1:
        return make()
    paths = SLAB_PATHS()
    which, sd = -1, None
    if rank == 0:
        sd = make()
        if os.environ.get("DM_BENCH_NO_SLAB", "0") in ("", "0"):
            for i, path in enumerate(paths):
                try:
                    synth.save_slab(sd, path)
                    which = i
                    break
                except OSError as ex:
                    print(f"bench.py: weight slab not written to {path} ({ex}); trying the next place", file=sys.stderr, flush=True)
                    synth.remove_slab(path)
    flag = torch.tensor([which], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.broadcast(flag, src=0)
    which = int(flag.item())
    _slab_in_use = paths[which] if which >= 0 else None
    if rank == 0:
        return sd
    return synth.load_slab(_slab_in_use) if _slab_in_use else make()


def drop_slab(rank: int) -> None:
    if rank == 0 and _slab_in_use:
        from diff_mining_amd import synth
        synth.remove_slab(_slab_in_use)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the single-image and side-workload legs (configs[3], [4]) behind the timed region")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the score-deviation leg (image 0 of the step re-scored in exact fp32 on the GPU by the fp32 net, after the timed region)")
    ap.add_argument("--images", type=int, default=N_IMG)
    ap.add_argument("--latent-dtype", choices=["f32", "f16"], default="f32",
                    help="dtype flow of add_noise / MSE: f32 = the reference's (compute.py:91-101), f16 = fp16 scheduler")
    ap.add_argument("--workload", choices=["typicality", "dift", "xray", "vae", "pixels"], default="typicality",
                    help="typicality = BASELINE configs[1]/[2] (the graded line); dift = configs[3]; xray = configs[4]; "
                         "vae = SURVEY 8f rank 2 (VAE encode of 8 images @512px); pixels = vae + typicality from images")
    ap.add_argument("--dift-dtype", choices=["f32", "f16"], default="f32",
                    help="--workload dift: f32 = the reference's arithmetic (dift.py:197-199: no torch_dtype, no autocast) on the fp32 "
                         "matrix cores (the gradable line); f16 = the fp16 engine (reduced precision, labelled so)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args))
        world = 1
    else:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a "
                     "number for a different GPU count than asked")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        # librccl prints a banner ("Librccl path : ...") on the C-level stdout (seen in tests/test_gpu_rccl.py on the GPU box; with a
        # pipe it is flushed at exit, i.e. AFTER the JSON line).  The contract is ONE JSON line on stdout: file descriptor 1 goes to
        # stderr for every native library, Python's sys.stdout keeps a private duplicate of the real stdout.
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        sys.stdout = os.fdopen(real_stdout, "w", buffering=1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not STUB:
            torch.cuda.set_device(0 if ONE_GPU else local_rank)
        dist.init_process_group("gloo" if (STUB or ONE_GPU) else "nccl", rank=rank, world_size=world)
    if ONE_GPU:
        local_rank = 0
    dev = torch.device("cpu") if STUB else torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if (STUB or ONE_GPU) else dev          # where the timing collectives' tensors live (gloo: host)
    sync = (lambda: None) if STUB else torch.cuda.synchronize

    from diff_mining_amd import synth
    from diff_mining_amd.typicality import gather_scores, shard_indices

    n_img = args.images
    per_img = N_DRAWS * N_COND
    slab_check = None
    if STUB:
        eng, sd = None, None
        if world > 1:
            # the weight hand-off of the real path on a stand-in dict: rank 0 writes the slab, the others map it; every rank's bytes
            # must be rank 0's (checked through the process group), and the slab must be gone afterwards
            fake = node_state_dict(rank, world, make=lambda: {f"w{i}": (np.arange(1000 + 37 * i, dtype=np.float32) * (i + 1)).astype(np.float16).reshape(-1, 1)
                                                              for i in range(5)})
            ck = torch.tensor([float(sum(int(np.asarray(v).view(np.uint16).astype(np.int64).sum()) for v in fake.values()))], dtype=torch.float64)
            allck = [torch.zeros_like(ck) for _ in range(world)]
            dist.all_gather(allck, ck)
            dist.barrier()
            used = _slab_in_use
            drop_slab(rank)
            slab_check = {"equal": all(a.item() == allck[0].item() for a in allck),
                          "mapped": bool(used is not None and (rank == 0 or not fake["w0"].flags.writeable)),
                          "removed": not any(os.path.exists(q) for q in SLAB_PATHS()), "slab": used}

        def step():                       # stands in for the engine: rank-dependent fake T(x|c), same gather path
            return gather_scores(torch.arange(n_img, dtype=torch.float32) + 100.0 * rank, n_img * world, rank, world)
    else:
        from diff_mining_amd.engine import UNetEngine, UNetEngineF32
        sd = node_state_dict(rank, world)
        eng = UNetEngineF32(local_rank) if (args.workload == "dift" and args.dift_dtype == "f32") else UNetEngine(local_rank)
        eng.load_state_dict(sd)
        if args.workload != "typicality":
            print(json.dumps(side_workload(args, eng, dev, sd)), flush=True)
            return
        ldt = torch.float32 if args.latent_dtype == "f32" else torch.float16
        x, eps, t, c = synth.synth_inputs(n_img * world, N_DRAWS, LAT, LAT,
                                          latent_dtype=np.float32 if args.latent_dtype == "f32" else np.float16)
        # this rank's images of the n_img * world work list: r::world, as the reference's `subs[i::sub_split]` (compute.py:339)
        # and as gather_scores files them
        x = torch.from_numpy(x)[shard_indices(n_img * world, rank, world)].to(dev)
        eps = torch.from_numpy(eps).to(dev)
        t = torch.from_numpy(t).to(dev)
        c = torch.from_numpy(c).to(dev)
        # the work list: image j under ITS OWN category prompt and the shared null prompt (compute.py:284-290, 182-192): 8 categories
        # (c[0] and seven more, same distribution) + c[1] = 9 distinct prompts, registered once
        gcat = torch.Generator().manual_seed(77)
        cats = torch.cat([c[:1].cpu(), torch.randn(n_img - 1, 77, 768, generator=gcat).to(torch.float16)]).to(dev)
        emb = torch.stack([torch.stack([cats[j], c[1]]) for j in range(n_img)]).contiguous()       # [n_img, 2, 77, 768]
        from diff_mining_amd.typicality import TypicalityScorer
        scorer = TypicalityScorer(eng, seed=42, N=N_DRAWS, t_min=0.1, t_max=0.7, latent_dtype=ldt)
        last = {"k": 0}
        # The reference hands every grid to the host (compute.py:156,160).  `value` keeps them on the device (the task's rule: a
        # PCIe-inclusive rate is never `value`); the leg behind the timed region measures the hand-over both ways — overlapped (the grid
        # of step k leaves on a copy stream into one of two pinned buffers while step k + 1 computes) and blocking (`.cpu()` per call).
        # DM_BENCH_D2H=1 puts the overlapped hand-over INSIDE the timed steps (same-box ABBA: +0.8 ms per step, of which +0.4 is the
        # cross-stream event alone: profiles/r06_ab_d2h_modes.txt)
        d2h = {"on": os.environ.get("DM_BENCH_D2H", "0") not in ("", "0")}
        copy_stream = torch.cuda.Stream(device=dev)
        host_grids = [torch.empty(n_img, N_DRAWS, N_COND, 4, LAT, LAT, dtype=torch.float16).pin_memory() for _ in range(2)]

        def step():
            # the product surface: D.compute_losses for the 8 images of the work-list slice in ONE engine call; the same N draws for
            # every image (manual_seed(42) precedes each image's draws, compute.py:139-141)
            grids = scorer.compute_losses_batch(x, emb, noises=eps, timesteps=t, to_host=False)    # [n_img, 10, 2, 4, 64, 64] fp16
            _, scores = eng.reduce_typicality_batched(grids, n_img, N_DRAWS, N_COND)               # one launch, no torch glue
            if d2h["on"]:
                copy_stream.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(copy_stream):
                    host_grids[last["k"] & 1].copy_(grids, non_blocking=True)
                grids.record_stream(copy_stream)         # the allocator must not hand the block to step k + 1 before the copy has read it
                last["k"] += 1
            last["grids"], last["loss"] = grids, scorer.last_loss32
            return gather_scores(scores, n_img * world, rank, world)     # world 1: the tensor itself (no kernel)

    if world > 1 and not STUB:
        dist.barrier()                     # every rank has copied its weights to its GPU: the node's slab in /dev/shm can go
        drop_slab(rank)
    if eng is not None and os.environ.get("DM_GRAPH", "0") not in ("", "0"):
        # hipGraph replay (diagnostic, with DM_BENCH_NOPROF=1: a replay carries no per-launch events): the legacy default
        # stream cannot be captured, so the steps run on a stream of their own
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(side)
    for _ in range(args.warmup):
        step()
    if eng is not None:
        eng.prof_enable(os.environ.get("DM_BENCH_NOPROF", "0") in ("", "0"))     # DM_BENCH_NOPROF=1: cost of the per-launch events (diagnostic)
        eng.prof_read()
    sampler = ClockSampler(local_rank) if (rank == 0 and not STUB) else None
    if world > 1:
        dist.barrier()
    sync()
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        all_scores = step()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    smi = sampler.stop() if sampler else {"sclk_mhz_mean": None, "power_w_mean": None, "clock_samples": 0, "clock_source": None,
                                           "clock_note": "stub run"}
    prof = eng.prof_read() if eng is not None else {"igemm_ms": 0.0, "igemm_flops": 0.0, "igemm_launches": 0,
                                                    "attn_ms": 0.0, "attn_flops": 0.0, "attn_launches": 0}
    if eng is not None:
        eng.prof_enable(False)
    tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
    rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        # every rank's own clock over the same barrier-bracketed region, so a straggler shows in the line; the reported time
        # is the MAX over ranks
        each = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(each, tt)
        rank_ms = [e.item() / args.steps * 1e3 for e in each]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = tt.item()

    # the collective on its own (outside the timed steps): the all-gather of the per-image scalars
    ag_ms = None
    if world > 1:
        probe = torch.zeros(n_img, dtype=torch.float32, device=dev)
        gather_scores(probe, n_img * world, rank, world)
        dist.barrier()
        sync()
        t1 = time.perf_counter()
        for _ in range(20):
            gather_scores(probe, n_img * world, rank, world)
        sync()
        ag_ms = (time.perf_counter() - t1) / 20 * 1e3

    # secondary: the fp16 grids handed to the host as the reference does (compute.py:156,160), outside `value`: overlapped on a copy
    # stream (what a pipelined caller gets), and blocking behind every call (the reference's own `.cpu()`); the last grid that reached
    # the host is compared with the one the device holds
    d2h_ms = {"overlapped": None, "blocking": None}
    d2h_equal = None
    d2h_in_value = False
    if eng is not None:
        d2h_in_value = d2h["on"]
    if eng is not None and world == 1 and not args.no_side:       # (--no-side: the profiler passes hold warm-up + timed steps only)
        was = d2h["on"]
        d2h["on"] = True
        step()
        sync()
        t1 = time.perf_counter()
        for _ in range(4):
            step()
        sync()
        d2h_ms["overlapped"] = (time.perf_counter() - t1) / 4 * 1e3
        d2h_equal = bool(torch.equal(host_grids[(last["k"] - 1) & 1], last["grids"].cpu()))
        d2h["on"] = False
        t1 = time.perf_counter()
        for _ in range(2):
            step()
            host_grids[0].copy_(last["grids"])
        sync()
        d2h_ms["blocking"] = (time.perf_counter() - t1) / 2 * 1e3
        d2h["on"] = was

    if rank == 0:
        total_images = n_img * world * args.steps
        value = total_images / dt
        # `achieved` / `frac` = the multiply-adds the launches ISSUED over their kernel time: a hardware roofline (the same
        # convention as the side workloads).  SURVEY 8d's ALGORITHMIC count (interpolate + 9 taps for Upsample2D.conv, of which
        # option up_fold executes 4/9: four 2x2 convolutions with pre-summed taps, igemm_pers_up.hip) is booked beside it as
        # `achieved_nominal` / `frac_nominal`
        folded = prof.get("igemm_flops_folded", 0.0)
        ig_tf_n = (prof["igemm_flops"] + folded) / (prof["igemm_ms"] * 1e-3) / 1e12 if prof["igemm_ms"] > 0 else 0.0
        ig_tf = prof["igemm_flops"] / (prof["igemm_ms"] * 1e-3) / 1e12 if prof["igemm_ms"] > 0 else 0.0
        at_tf = prof["attn_flops"] / (prof["attn_ms"] * 1e-3) / 1e12 if prof["attn_ms"] > 0 else 0.0
        # executed work = what the engine's launches actually computed on this rank (shared-draw prefix and the cached
        # cross-attention K/V are NOT re-done per prompt); nominal = 803.27 GFLOP x forwards, as SURVEY §8d counts
        executed = (prof["igemm_flops"] + prof["attn_flops"]) / max(args.steps, 1)
        nominal = n_img * per_img * FLOP_PER_FORWARD_64
        step_s = dt / args.steps
        out = {
            "metric": "typicality-scored images/sec/node (512px, 10 t×2 prompts)",
            "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "configs[1]: SDv1.5 U-Net fp16, 512x512 (latent 64x64), 10 t-samples x 2 prompts, "
                                   f"batch {n_img} images/GPU = {n_img * per_img} U-Net forwards/step/GPU, synthetic weights",
                       "entry": "TypicalityScorer.compute_losses_batch (dm_score_conds_slots): each image under its own category + the shared null prompt",
                       "images_per_gpu_per_step": n_img, "unet_forwards_per_image": per_img,
                       "latent_dtype_flow": args.latent_dtype,
                       "parallelism": f"image-sharded x{world}, one all-gather of T(x|c)"},
            "roofline": {"bound": "mfma", "kernel": "igemm family: igemm_pers_kernel / igemm_pers_tr_kernel (persistent 256x320 tile; tr = 3x3 convolutions with horizontal tap reuse) + igemm_kernel (128x320 / 128x160 tile) incl. their LayerNorm-folded and split-K instantiations (implicit-GEMM conv3x3/1x1/linear)",
                         "achieved": round(ig_tf, 2), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ig_tf / PEAK_TFLOPS, 4),
                         "achieved_executed": round(ig_tf, 2), "frac_executed": round(ig_tf / PEAK_TFLOPS, 4),
                         "achieved_nominal": round(ig_tf_n, 2), "frac_nominal": round(ig_tf_n / PEAK_TFLOPS, 4),
                         "flops_convention": "achieved (= achieved_executed) = multiply-adds issued by the launches / their kernel time; achieved_nominal = "
                                             "SURVEY 8d's algorithmic count (Upsample2D.conv = interpolate + 9 taps; up_fold issues 4/9 of it)",
                         "folded_tflop_per_step": round(folded / max(args.steps, 1) / 1e12, 3),
                         "traffic": hbm_traffic_per_launch(prof["igemm_launches"] / max(args.steps, 1)),
                         "launches": prof["igemm_launches"], "kernel_ms_total": round(prof["igemm_ms"], 3),
                         "whole_path_tflops": round(executed / step_s / 1e12, 2),
                         "whole_path_frac": round(executed / step_s / 1e12 / PEAK_TFLOPS, 4),
                         "executed_tflop_per_step": round(executed / 1e12, 3),
                         "nominal_tflop_per_step": round(nominal / 1e12, 3),
                         "whole_path_tflops_nominal": round(nominal / step_s / 1e12, 2),
                         "attention_tflops": round(at_tf, 2), "attention_ms_total": round(prof["attn_ms"], 3),
                         **clock_normalised(smi, ig_tf, PEAK_TFLOPS),
                         "whole_path_frac_at_sustained_clock": (round(executed / step_s / 1e12 / (PEAK_TFLOPS * smi["sclk_mhz_mean"] / NOMINAL_SCLK_MHZ), 4)
                                                                if smi.get("sclk_mhz_mean") else None)},
            "allgather_ms": None if ag_ms is None else round(ag_ms, 4),
            "grid_d2h": ("stub run" if eng is None else
                         {"in_value": bool(d2h_in_value),
                          "ms_per_step_with_overlapped_d2h": None if d2h_ms["overlapped"] is None else round(d2h_ms["overlapped"], 3),
                          "images_per_s_with_overlapped_d2h": None if d2h_ms["overlapped"] is None else round(n_img / d2h_ms["overlapped"] * 1e3, 4),
                          "ms_per_step_with_blocking_d2h": None if d2h_ms["blocking"] is None else round(d2h_ms["blocking"], 3),
                          "images_per_s_with_blocking_d2h": None if d2h_ms["blocking"] is None else round(n_img / d2h_ms["blocking"] * 1e3, 4),
                          "last_host_grid_equals_device": d2h_equal,
                          "bytes_per_step": n_img * N_DRAWS * N_COND * 4 * LAT * LAT * 2}),
            "scores_checksum": float(all_scores.double().sum().item()),
        }
        if eng is not None and not args.no_side:
            # (skipped with --no-side, like the other legs behind the timed region: the rocprofv3 runs of tools/profile_round.sh use it, so the
            # kernel-stats summary holds the step's kernels only)
            # the matrix cores' own ceiling on this box, measured right behind the timed steps (the chip is warm and at its power cap):
            # the igemm tile's MFMA stream with nothing else in the loop, random fp16 operands (~60 ms) and zeros (what a zero-filled
            # benchmark would see).  `frac` prices against the nominal 2.5 PFLOP/s; `frac_of_measured_mfma_rate` against this.
            try:
                smp = ClockSampler(local_rank, period_s=0.01)
                torch.cuda.synchronize()
                smp.start()
                pk = eng.measure_mfma_rate(80000, False)                      # ~115 ms: long enough for the clock to settle and be sampled
                ck = smp.stop()
                pz = eng.measure_mfma_rate(20000, True)
                out["roofline"].update({"mfma_only_tflops_measured": round(pk["tflops"], 1),
                                        "mfma_only_sclk_mhz": ck.get("sclk_mhz_mean"), "mfma_only_power_w": ck.get("power_w_mean"),
                                        "mfma_only_tflops_zero_operands": round(pz["tflops"], 1),
                                        "frac_of_measured_mfma_rate": round(ig_tf / pk["tflops"], 4),
                                        "whole_path_frac_of_measured_mfma_rate": round(executed / step_s / 1e12 / pk["tflops"], 4)})
            except Exception as ex:                            # noqa: BLE001  (an optional measurement must not cost the line)
                out["roofline"]["mfma_only_tflops_measured"] = None
                out["roofline"]["mfma_only_note"] = f"{type(ex).__name__}: {ex}"
        if eng is not None:
            out["engine_stats"] = eng.stats()
            from diff_mining_amd.engine import get_options
            # the algebraic rewrites / schedules the line was measured with (all numerically equivalent to the layer-by-layer order)
            out["config"]["engine_options"] = get_options()
        out["roofline"]["traffic_recorded_from"] = TRAFFIC_SOURCE.get("file")
        if world > 1:
            # every rank's clock arrived through the all-gather above: a line for N GPUs is printed only if N ranks measured
            if len(rank_ms) != args.gpus or dist.get_world_size() != args.gpus:
                sys.exit(f"bench.py: {len(rank_ms)} rank timings / world size {dist.get_world_size()} for --gpus {args.gpus}; no line")
            out["ranks_seen"] = len(rank_ms)
            out["backend"] = dist.get_backend()
            try:
                out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if not STUB else None
            except Exception as ex:                          # a missing version string must not cost the measurement
                out["rccl_version"] = f"unavailable ({type(ex).__name__})"
            out["rank_ms_per_step"] = {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3),
                                       "all": [round(v, 3) for v in rank_ms]}
        if ONE_GPU and world > 1:
            out["rehearsal"] = (f"{world} ranks with real engines on ONE GPU over gloo (DM_BENCH_ONE_GPU=1): a wiring check of the N > 1 path, "
                                "NOT a scaling number — the ranks share the device")
        if STUB:
            out["data"] = "stub (DM_BENCH_STUB=1: launcher / gather path only, no engine)"
            out["weight_slab_check"] = slab_check
        net32 = None
        if world == 1 and not STUB and (not args.no_parity or not args.no_side):
            from diff_mining_amd.engine import UNetEngineF32
            net32 = UNetEngineF32(dev.index or 0)          # the fp32 net: exact-fp32 reference of the deviation leg and the DIFT side workload
            net32.load_state_dict(sd)
        if not args.no_parity and world == 1 and not STUB:
            out["score_deviation"] = score_deviation(net32, x[:1], eps, t, emb[0], last["loss"], n_img)
        if not args.no_side and world == 1 and not STUB:
            out["single_image_call"] = single_image_call(eng, x[:1], emb[0], ldt)
            out["side_workloads"] = {}
            for w, e_ in (("dift", net32), ("xray", eng)):
                a2 = argparse.Namespace(**vars(args))
                a2.workload, a2.dift_dtype, a2.steps, a2.warmup, a2.no_cpu_baseline = w, "f32", 3, 1, True
                ln = side_workload(a2, e_, dev, sd)
                out["side_workloads"][w] = {"metric": ln["metric"], "value": ln["value"], "unit": ln["unit"], "ms_per_step": ln["ms_per_step"],
                                            "steps": 3, "dtype": ln["dtype"], "config": ln["config"]["workload"],
                                            "roofline": {k: ln["roofline"].get(k) for k in ("kernel", "achieved", "peak", "frac", "attention_tflops",
                                                                                           "whole_path_frac_nominal", "sclk_mhz_mean", "power_w_mean",
                                                                                           "frac_at_sustained_clock")}}
        if net32 is not None:
            net32.close()
        if not args.no_cpu_baseline and world == 1 and not STUB:
            out["cpu_baseline"] = cpu_baseline(sd)
        print(json.dumps(out), flush=True)
        # north_star: "<= 1e-3 score deviation from reference" — a line whose own deviation leg exceeds it is printed and FAILS
        dev_bad = "score_deviation" in out and out["score_deviation"]["loss_grid_rel_l2"] > SCORE_DEVIATION_MAX
    else:
        dev_bad = False
    if world > 1:
        dist.destroy_process_group()
    if dev_bad:
        sys.exit(f"bench.py: score_deviation.loss_grid_rel_l2 exceeds {SCORE_DEVIATION_MAX:g}")


def single_image_call(eng, x0, embeds, ldt):
    """The reference's own call shape: `D.compute_losses` on ONE image (compute.py:134-160) = `TypicalityScorer.compute_losses`, at
    N = 10 draws (one engine call of 20 U-Net samples: the 8x8 / 16x16 / 32x32 levels give the 256x320 tile 20 / 80 / 160 tiles for
    256 CUs) and at the reference default N = 100 (compute.py:106: 200 samples, engine chunks of 160 + 40)."""
    from diff_mining_amd.typicality import TypicalityScorer
    res = {}
    for N, reps in ((10, 5), (100, 2)):
        sc = TypicalityScorer(eng, seed=42, N=N, t_min=0.1, t_max=0.7, latent_dtype=ldt)
        noises, ts = sc.draw(x0.shape)
        noises, ts = noises.to(eng.device), ts.to(eng.device)
        sc.compute_losses(x0, embeds, noises=noises, timesteps=ts, to_host=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g = sc.compute_losses(x0, embeds, noises=noises, timesteps=ts, to_host=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res[f"N{N}"] = {"images_per_s": round(1.0 / dt, 3), "ms_per_image": round(dt * 1e3, 2), "unet_samples_per_image": 2 * N,
                        "images_per_s_at_10_draws_equivalent": round(N / 10.0 / dt, 3), "grid": list(g.shape)}
    res["note"] = ("one image per engine call, scores left on the device; the batched entry (`value`) rides 8 images per call. "
                   "N100 per 10 draws compares with `value`: the same engine batch (160 samples), one image's prompts")
    return res


def score_deviation(net, x0, eps, t, c, loss, n_img):
    """north_star ends on "<= 1e-3 score deviation from reference": image 0 of the timed step (its 10 draws x 2 prompts, the very
    losses the step produced) against the EXACT-fp32 evaluation of the same U-Net on the same inputs — the fp32 net (dm_f32_*,
    product code, 2-4e-6 from the CPU oracle's autocast=False arithmetic: tests/test_gpu_f32.py): dm_f32_score = compute.py:95-102 with no
    autocast.  Outside the timed region; the CPU oracle cannot reach this size."""
    t0 = time.perf_counter()
    net.set_prompts(c.float())                                    # image 0's own (category, null) pair
    ref = net.score_conds(x0, eps, t, N_COND)                     # dm_f32_score: fp32 add_noise -> U-Net -> squared error, cond-major rows
    ref = ref.view(N_COND, N_DRAWS, 4, LAT, LAT).transpose(0, 1)
    got = loss.view(N_COND, n_img, N_DRAWS, 4, LAT, LAT)[:, 0].transpose(0, 1).float()      # [N,2,4,h,w] of image 0 (cond 0 = c, 1 = null)
    T = (got[:, 1] - got[:, 0]).double().mean().item()
    T32 = (ref[:, 1] - ref[:, 0]).double().mean().item()
    ml = ref.double().mean().item()
    return {"reference": "exact-fp32 evaluation of the same U-Net on the same (x, t, eps, c) on the GPU (fp32 net), image 0 of the step: "
                         "10 draws x 2 prompts @64x64",
            "loss_grid_rel_l2": round(((got - ref).double().norm() / ref.double().norm()).item(), 7),
            "T_engine": T, "T_fp32": T32, "abs_dT_over_abs_T": round(abs(T - T32) / abs(T32), 7),
            "abs_dT_over_mean_loss": round(abs(T - T32) / ml, 9), "seconds": round(time.perf_counter() - t0, 1)}


TRAFFIC_SOURCE = {}


def hbm_traffic_per_launch(launches_per_step):
    """HBM bytes per igemm launch (one launch_igemm call = one GEMM of the network; the head / tail row split makes some
    of them two kernel dispatches) from the PMC passes committed under profiles/ (FETCH_SIZE doubled per the gfx950
    correction, + WRITE_SIZE): family bytes per step / launches per step.  rocprofv3 cannot run inside this process,
    so the number is the recorded one for this kernel build, or None when no record exists."""
    for tag in ("r06_end", "r05_end", "r05_final", "r04_final", "r03_final", "r02_final"):          # the newest record that is committed
        try:
            with open(os.path.join(ROOT, "profiles", tag + "_pmc.json")) as f:
                k = json.load(f)["kernels"]["igemm_family"]
            # a RECORDED value (separate rocprofv3 --pmc passes of this command), not measured in this run: the line says which file
            TRAFFIC_SOURCE["file"] = f"profiles/{tag}_pmc.json"
            return round((k["fetch_GB_per_step"] + k["write_GB_per_step"]) * 1e9 / launches_per_step)
        except Exception:
            continue
    return None


def side_workload(args, eng, dev, sd):
    """BASELINE configs[3] (DIFT-161 tap, batch 64) and configs[4] (1024 px per-pixel heat-map) — and the 8f extras
    (VAE encode, scoring from pixels) — with the same keys as the graded line: `roofline` from the engine's live per-launch
    HIP events over the timed steps (igemm family), `cpu_baseline` from the oracle on a bounded sample."""
    from diff_mining_amd import synth
    cfg, dtype_note = None, None
    f32 = args.workload == "dift" and args.dift_dtype == "f32"
    if args.workload == "dift":
        n_lat, ens, lat = 8, 8, 64
        x, eps, _, c = synth.synth_inputs(n_lat, ens, lat, lat)
        xt = torch.from_numpy(x).float().to(dev).repeat_interleave(ens, 0)
        et = torch.from_numpy(eps).float().to(dev).repeat(n_lat, 1, 1, 1)
        a = 0.81210744                                          # acp[161]
        noisy = ((a ** 0.5) * xt + ((1 - a) ** 0.5) * et)
        noisy = noisy if f32 else noisy.half()
        eng.set_prompts(torch.from_numpy(c[:1]).to(dev))
        slots = torch.zeros(n_lat * ens, dtype=torch.int32, device=dev)
        tt = torch.tensor(161, device=dev)

        def step():
            return eng.dift(noisy, tt, slots, 1, ens)[1]
        units, name, flop = n_lat, "DIFT-161 images/s (ensemble 8, tap up_blocks[1], batch 64 @64x64 latent)", 438.79e9 * ens
        cfg = {"workload": "configs[3]: DIFT-161 feature extraction, single-timestep U-Net forward with the up_blocks[1] tap "
                           "(dift.py:133-165; BASELINE.json says 'mid-block': SURVEY F6a), batch 64 = 8 images x ensemble 8, 64x64 latent",
               "images_per_step": n_lat, "ensemble": ens, "t": 161, "up_ft_index": 1}
        dtype_note = None if f32 else ("the engine computes this path in fp16 (fp32 accumulation / norms / softmax), the reference's SDFeaturizer runs "
                      "the U-Net in fp32 (dift.py:197-199): a REDUCED-PRECISION number by the bench rule; descriptor deviation vs the "
                      "fp32 oracle: cosine >= 1 - 5e-7 (tests/test_gpu_e2e.py::test_dift_descriptor_deviation_vs_fp32_oracle)")
    elif args.workload in ("vae", "pixels"):
        n_img, lat = N_IMG, LAT
        eng.load_vae_state_dict(synth.synth_vae_state_dict(seed=0, dtype=np.float16))
        img = torch.from_numpy(synth.synth_image(n_img, lat * 8, lat * 8)).to(dev)
        _, eps, t, c = synth.synth_inputs(n_img, N_DRAWS, lat, lat)
        vnoise = torch.from_numpy(synth.synth_inputs(n_img, 1, lat, lat)[0]).to(dev)
        eu, tu = torch.from_numpy(eps).to(dev).repeat(n_img, 1, 1, 1), torch.from_numpy(t).to(dev).repeat(n_img)
        xi = torch.arange(n_img, dtype=torch.int32, device=dev).repeat_interleave(N_DRAWS)
        eng.set_prompts(torch.from_numpy(c).to(dev))
        vae_flop = 1116.66e9                                     # encoder @512x512, 2 FLOP/MAC incl. attention
        if args.workload == "vae":
            def step():
                return eng.vae_encode(img, vnoise)
            units, name, flop = n_img, "VAE-encoded images/s (SDv1.5 AutoencoderKL encoder + posterior sample, 512x512, batch 8)", vae_flop
        else:
            def step():
                x = eng.vae_encode(img, vnoise)
                loss = eng.score_conds(x, eu, tu, N_COND, xi)
                return eng.reduce_typicality_batched(loss, n_img, N_DRAWS, N_COND, cond_major=True)[1]
            units, name, flop = n_img, ("typicality-scored images/s from pixels (VAE encode + 10 t x 2 prompts, 512px, "
                                        "batch 8)"), vae_flop + 20 * 803.27e9
        cfg = {"workload": f"SURVEY 8f rank 2 ({args.workload}): 8 images @512x512", "images_per_step": n_img}
    else:
        lat = 128
        x, eps, t, c = synth.synth_inputs(1, N_DRAWS, lat, lat, latent_dtype=np.float32)
        xd, ed, td, cd = (torch.from_numpy(a).to(dev) for a in (x, eps, t, c))
        eng.set_prompts(cd)

        def step():
            loss = eng.score_conds(xd, ed, td, N_COND, latent_dtype=torch.float32)         # cond-major rows, as compute.py:150-155
            return eng.reduce_typicality_batched(loss, 1, N_DRAWS, N_COND, cond_major=True)[0][0]     # [128,128] E_N[L_null - L_c]
        units, name, flop = 1, "X-ray 1024x1024 per-pixel typicality heat-maps/s (latent 128x128, 10 t x 2 prompts)", 4674.01e9 * 20
        cfg = {"workload": "configs[4]: X-ray 1024x1024 per-pixel typicality heat-map (no patch reduction), latent 128x128, "
                           "10 t-samples x 2 prompts = 20 U-Net forwards per heat-map (applications/xray/compute.py:117-143,210-218)",
               "heatmaps_per_step": 1, "unet_forwards_per_heatmap": 20, "latent_dtype_flow": "f32"}
    for _ in range(args.warmup):
        step()
    eng.prof_enable(os.environ.get("DM_BENCH_NOPROF", "0") in ("", "0"))
    eng.prof_read()
    sampler = ClockSampler(dev.index or 0)
    torch.cuda.synchronize()
    sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    smi = sampler.stop()
    prof = eng.prof_read()
    eng.prof_enable(False)
    val = units * args.steps / dt
    ig_tf = prof["igemm_flops"] / (prof["igemm_ms"] * 1e-3) / 1e12 if prof["igemm_ms"] > 0 else 0.0
    at_tf = prof["attn_flops"] / (prof["attn_ms"] * 1e-3) / 1e12 if prof["attn_ms"] > 0 else 0.0
    # fp32 DIFT: the fp32 matrix cores (v_mfma_f32_16x16x4_f32), 256 FLOP/clk/CU = 157.3 TFLOP/s (MI355X_MICROARCH.md: "Peak FP32 (matrix)")
    peak = PEAK_TFLOPS_F32 if f32 else PEAK_TFLOPS
    line = {"metric": name, "value": round(val, 4), "unit": "images/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if f32 else "f16", "data": "synthetic", "config": cfg,
            "roofline": {"bound": "mfma", "kernel": "gemm32_kernel (fp32 implicit GEMM, 128x160x32 tile, v_mfma_f32_32x32x2_f32)" if f32 else
                         "igemm family (igemm_pers_kernel + igemm_kernel)", "achieved": round(ig_tf, 2),
                         "peak": peak, "unit": "TFLOP/s", "frac": round(ig_tf / peak, 4), "traffic": None,
                         "launches": prof["igemm_launches"], "kernel_ms_total": round(prof["igemm_ms"], 3),
                         "attention_tflops": round(at_tf, 2), "attention_ms_total": round(prof["attn_ms"], 3),
                         "whole_path_tflops_nominal": round(val * flop / 1e12, 2),
                         "whole_path_frac_nominal": round(val * flop / 1e12 / peak, 4),
                         **clock_normalised(smi, ig_tf, peak)},
            "out_shape": list(out.shape), "memory": eng.memory()}
    if args.workload == "dift":
        line["reference_dtype"] = "f32"
    if dtype_note:
        line["dtype_note"] = dtype_note
    if not args.no_cpu_baseline and args.workload in ("dift", "xray"):
        line["cpu_baseline"] = cpu_baseline_side(sd, args.workload)
    return line


def _oracle_threads():
    """The fp32 oracle peaks at 16 threads on the GPU box's host (measured 681 / 500 / 169 / 78 GFLOP/s at 16 / 32 / 64 / 128
    threads), so 16 are used and reported as `cores`."""
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    return cores


def cpu_baseline(sd, forwards=None):
    """Oracle (fp32 PyTorch CPU, kind "port") on ONE FULL IMAGE of the workload: 10 draws x 2 prompts of one 64x64 latent =
    the 20 U-Net forwards D.compute_losses makes per image (compute.py:145-152), in two reference-sized chunks of B = 5
    draws (U-Net batch 10).  DM_CPU_BASELINE_FORWARDS=4 gives the r02 quick sample (2 draws x 2 prompts)."""
    from diff_mining_amd import synth
    from oracle import unet_ref as R
    cores = _oracle_threads()
    forwards = forwards or int(os.environ.get("DM_CPU_BASELINE_FORWARDS", "20"))
    draws = max(1, forwards // N_COND)
    sdt = {k: torch.from_numpy(v).float() for k, v in sd.items()}
    x, eps, t, c = (torch.from_numpy(a) for a in synth.synth_inputs(1, draws, LAT, LAT))
    with torch.no_grad():
        R.compute_loss(sdt, x[:, :, :16, :16], eps[:1, :, :16, :16], t[:1], c[:1].float(), autocast=False)     # warm-up (small)
        t0 = time.perf_counter()
        grid = R.compute_losses(sdt, x.float(), c.float(), eps.float(), t, B=5, autocast=False)
        dt = time.perf_counter() - t0
    forwards = draws * N_COND
    return {"value": round(forwards / (N_DRAWS * N_COND) / dt, 6), "unit": "images/s", "cores": cores, "kind": "port",
            "kind_detail": "port: the fp32 oracle restatement (oracle/unet_ref.py) — not diffusers, which is absent from the image",
            "sample": f"{forwards} U-Net forwards @64x64 ({'one full image: 10 draws x 2 prompts' if forwards == 20 else f'{forwards}/20 of one image'}) "
                      f"in {dt:.1f}s, fp32 oracle, torch threads={cores}, grid {list(grid.shape)}",
            "gflops": round(forwards * FLOP_PER_FORWARD_64 / dt / 1e9, 1)}


def cpu_baseline_side(sd, workload):
    """The oracle on a bounded sample of a side workload: DIFT = one image (ensemble 8 @64x64, fp32 like the reference,
    dift.py:197-199); X-ray = one draw x 2 prompts @128x128 (2 of the 20 forwards of one heat-map)."""
    from diff_mining_amd import synth
    from oracle import unet_ref as R
    cores = _oracle_threads()
    sdt = {k: torch.from_numpy(v).float() for k, v in sd.items()}
    with torch.no_grad():
        if workload == "dift":
            x, eps, _, c = (torch.from_numpy(a) for a in synth.synth_inputs(1, 8, LAT, LAT))
            noisy = R.add_noise(x.float().expand(8, -1, -1, -1), eps.float(), torch.tensor(161))
            t0 = time.perf_counter()
            R.dift_features(sdt, noisy, 161, c[:1].float().expand(8, -1, -1), 1)
            dt = time.perf_counter() - t0
            return {"value": round(1.0 / dt, 6), "unit": "images/s", "cores": cores, "kind": "port",
                    "sample": f"one image = ensemble 8 @64x64, tap up_blocks[1], in {dt:.1f}s, fp32 oracle, torch threads={cores}",
                    "gflops": round(8 * 438.79 / dt, 1)}
        x, eps, t, c = (torch.from_numpy(a) for a in synth.synth_inputs(1, 1, 128, 128))
        nb, tb = torch.cat([eps] * 2).float(), torch.cat([t] * 2)
        t0 = time.perf_counter()
        R.compute_loss(sdt, x.float(), nb, tb, c.float(), autocast=False)
        dt = time.perf_counter() - t0
        return {"value": round(2.0 / 20.0 / dt, 6), "unit": "images/s", "cores": cores, "kind": "port",
                "sample": f"1 draw x 2 prompts @128x128 (2 of the 20 forwards of one heat-map) in {dt:.1f}s, fp32 oracle, torch threads={cores}",
                "gflops": round(2 * 4674.01 / dt, 1)}


if __name__ == "__main__":
    main()
