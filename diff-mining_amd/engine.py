"""ctypes binding of libdm_engine.so (C ABI: include/dm_engine.h) — PyTorch-ROCm is used only for
device memory, streams and `torch.distributed`; every FLOP of the U-Net runs in the HIP library.

There is NO fallback path: if the library is missing or no GPU is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdm_engine.so")
_lib = None
_CHECK_DEVICE_SLOTS = os.environ.get("DM_CHECK_SLOTS", "0") not in ("", "0")     # debug: validate prompt slots that live on the GPU

# every symbol include/dm_engine.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "dm_version", "dm_scheduler_alphas_cumprod", "dm_timestep_sinusoid", "dm_engine_create",
    "dm_engine_destroy", "dm_last_error", "dm_engine_load_weight", "dm_engine_finalize",
    "dm_engine_set_prompts", "dm_score", "dm_score_conds", "dm_score_conds_slots", "dm_unet_forward", "dm_dift", "dm_dift_shape",
    "dm_reduce_typicality", "dm_typicality_image", "dm_prof_enable", "dm_prof_read", "dm_engine_memory",
    "dm_op_igemm", "dm_op_attention", "dm_op_groupnorm", "dm_op_layernorm",
    "dm_op_conv_temb_gn_blocks", "dm_op_gn_blocks", "dm_op_groupnorm_blocks", "dm_op_conv_out",
    "dm_engine_load_vae_weight", "dm_engine_finalize_vae", "dm_vae_encode", "dm_op_attention512", "dm_patch_embed",
    "dm_engine_load_clip_weight", "dm_engine_finalize_clip", "dm_clip_encode", "dm_op_igemm_splitk",
    "dm_op_ln_stats", "dm_op_igemm_ln", "dm_reduce_typicality_batched", "dm_op_igemm_tile", "dm_op_igemm_head_rows", "dm_set_option",
    "dm_engine_reserve", "dm_engine_stats", "dm_op_groupnorm_conv1x1", "dm_op_igemm_shortcut", "dm_normalize_map",
    "dm_op_fold_upconv_weights", "dm_op_upconv_folded", "dm_prof_read_folded", "dm_get_option", "dm_measure_mfma_rate",
    "dm_f32_create", "dm_f32_destroy", "dm_f32_last_error", "dm_f32_load_weight", "dm_f32_finalize", "dm_f32_set_prompts",
    "dm_f32_unet_forward", "dm_f32_dift", "dm_f32_prof_enable", "dm_f32_prof_read", "dm_f32_memory", "dm_f32_op_gemm",
    "dm_f32_op_attention", "dm_f32_op_groupnorm", "dm_f32_op_layernorm", "dm_f32_load_vae_weight", "dm_f32_finalize_vae",
    "dm_f32_vae_encode", "dm_f32_score", "dm_f32_load_clip_weight", "dm_f32_finalize_clip", "dm_f32_clip_encode",
]


def get_options(names=("ln_fold", "gn_fold", "ff_fold", "sc_fold", "up_fold", "tap_reuse", "ln_inkernel", "igemm_splitk", "q_once", "gn_epi", "conv_out_rows", "gn_skip", "attn_pipe", "graph")) -> dict:
    """Current values of the library's runtime switches (dm_get_option); {} with a library that predates the getter."""
    lib = load_library()
    out = {}
    if not hasattr(lib, "dm_get_option"):
        return out
    for n in names:
        v = C.c_int32()
        if lib.dm_get_option(n.encode(), C.byref(v)) == 0:
            out[n] = v.value
    return out


class EngineError(RuntimeError):
    pass


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load libdm_engine.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("DM_ENGINE_LIB") or LIB_PATH   # DM_ENGINE_LIB: kernel A/B experiments
    if not os.path.exists(p):
        raise EngineError(
            f"{p} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "The typicality engine has no CPU / PyTorch fallback.")
    # PyTorch-ROCm ships its own libamdhip64; importing torch first makes this library bind to that
    # same HIP runtime (two runtimes in one process cannot see each other's device context).
    import torch  # noqa: F401
    lib = C.CDLL(p)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.dm_version.restype = C.c_char_p
    lib.dm_last_error.restype = C.c_char_p
    lib.dm_last_error.argtypes = [vp]
    lib.dm_scheduler_alphas_cumprod.argtypes = [i32, C.c_float, C.c_float, C.POINTER(C.c_float)]
    lib.dm_timestep_sinusoid.argtypes = [i32, i32, C.POINTER(C.c_float)]
    lib.dm_engine_create.argtypes = [i32, C.POINTER(vp)]
    lib.dm_engine_destroy.argtypes = [vp]
    lib.dm_engine_destroy.restype = None
    lib.dm_engine_load_weight.argtypes = [vp, C.c_char_p, vp, i32, C.POINTER(i64), i32]
    lib.dm_engine_finalize.argtypes = [vp]
    lib.dm_engine_set_prompts.argtypes = [vp, vp, i32, vp]
    lib.dm_score.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.dm_score_conds.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.dm_score_conds_slots.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.dm_reduce_typicality_batched.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.dm_unet_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
    lib.dm_dift.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.dm_dift_shape.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.dm_reduce_typicality.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.dm_typicality_image.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.dm_prof_enable.argtypes = [vp, i32]
    lib.dm_measure_mfma_rate.argtypes = [vp, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if hasattr(lib, "dm_prof_read_folded"):
        lib.dm_prof_read_folded.argtypes = [vp, C.POINTER(C.c_double)]
    lib.dm_prof_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i64),
                                 C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i64)]
    lib.dm_engine_memory.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.dm_op_igemm.argtypes = [vp] * 8 + [i32] * 11
    lib.dm_op_attention.argtypes = [vp] * 5 + [i32] * 4 + [i64] * 4 + [vp] + [i32] * 5 + [C.c_float]
    lib.dm_op_groupnorm.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, C.c_float, vp, vp, i32, vp]
    if hasattr(lib, "dm_op_conv_out"):
        lib.dm_op_conv_out.argtypes = [vp] * 5 + [i32] * 4 + [vp, vp]
    if hasattr(lib, "dm_op_gn_blocks"):          # (absent only from older A/B libraries loaded through DM_ENGINE_LIB)
        lib.dm_op_conv_temb_gn_blocks.argtypes = [vp] * 6 + [i32] * 6 + [vp, C.POINTER(C.c_int)]
        lib.dm_op_gn_blocks.argtypes = [vp, vp, i32, i32, i32, vp]
        lib.dm_op_groupnorm_blocks.argtypes = [vp, vp, vp, i32, i32, i32, i32, C.c_float, vp, vp, i32, vp]
    lib.dm_op_layernorm.argtypes = [vp, vp, i32, i32, vp, vp, C.c_float, vp]
    lib.dm_engine_load_vae_weight.argtypes = [vp, C.c_char_p, vp, i32, C.POINTER(i64), i32]
    lib.dm_engine_finalize_vae.argtypes = [vp]
    lib.dm_vae_encode.argtypes = [vp, vp, vp, i32, i32, i32, i32, C.c_float, vp, vp, vp, vp]
    lib.dm_patch_embed.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp]
    lib.dm_op_ln_stats.argtypes = [vp, vp, i32, i32, C.c_float, vp]
    lib.dm_op_igemm_ln.argtypes = [vp] * 7 + [i32] * 4
    lib.dm_op_igemm_splitk.argtypes = [vp] * 8 + [i32] * 11 + [vp]
    lib.dm_engine_load_clip_weight.argtypes = [vp, C.c_char_p, vp, i32, C.POINTER(i64), i32]
    lib.dm_engine_finalize_clip.argtypes = [vp]
    lib.dm_clip_encode.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    lib.dm_op_igemm_tile.argtypes = [i32, i32, i32, i32]
    lib.dm_set_option.argtypes = [C.c_char_p, i32]
    if hasattr(lib, "dm_get_option"):
        lib.dm_get_option.argtypes = [C.c_char_p, C.POINTER(i32)]
    lib.dm_op_attention512.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, C.c_float]
    if hasattr(lib, "dm_op_igemm_shortcut"):
        lib.dm_op_igemm_shortcut.argtypes = [vp] * 8 + [i32] * 8
    if hasattr(lib, "dm_op_upconv_folded"):
        lib.dm_op_fold_upconv_weights.argtypes = [vp, i32, i32, vp]
        lib.dm_op_upconv_folded.argtypes = [vp] * 5 + [i32] * 5
    if hasattr(lib, "dm_op_groupnorm_conv1x1"):
        lib.dm_op_groupnorm_conv1x1.argtypes = [vp, vp, i32, i32, i32, i32, C.c_float, vp, vp, vp, vp, i32, vp]
    if hasattr(lib, "dm_engine_reserve"):        # absent only from older A/B libraries loaded through DM_ENGINE_LIB
        lib.dm_engine_reserve.argtypes = [vp, i32, i32, i32, i32, i32, vp]
        lib.dm_engine_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    if hasattr(lib, "dm_normalize_map"):
        lib.dm_normalize_map.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp]
    if hasattr(lib, "dm_f32_create"):            # the fp32 U-Net (DIFT arithmetic), r04
        lib.dm_f32_create.argtypes = [i32, C.POINTER(vp)]
        lib.dm_f32_destroy.argtypes = [vp]
        lib.dm_f32_destroy.restype = None
        lib.dm_f32_last_error.restype = C.c_char_p
        lib.dm_f32_last_error.argtypes = [vp]
        lib.dm_f32_load_weight.argtypes = [vp, C.c_char_p, vp, i32, C.POINTER(i64), i32]
        lib.dm_f32_finalize.argtypes = [vp]
        lib.dm_f32_set_prompts.argtypes = [vp, vp, i32, vp]
        lib.dm_f32_unet_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
        lib.dm_f32_dift.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, vp]
        lib.dm_f32_prof_enable.argtypes = [vp, i32]
        lib.dm_f32_prof_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i64),
                                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i64)]
        lib.dm_f32_memory.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        lib.dm_f32_op_gemm.argtypes = [vp] * 8 + [i32] * 10
        lib.dm_f32_op_attention.argtypes = [vp] * 5 + [i32] * 4 + [i64] * 4 + [vp] + [i32] * 6 + [C.c_float]
        lib.dm_f32_op_groupnorm.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, C.c_float, vp, vp, i32, vp, vp]
        lib.dm_f32_op_layernorm.argtypes = [vp, vp, i32, i32, vp, vp, C.c_float, vp]
        lib.dm_f32_load_vae_weight.argtypes = [vp, C.c_char_p, vp, i32, C.POINTER(i64), i32]
        lib.dm_f32_finalize_vae.argtypes = [vp]
        lib.dm_f32_vae_encode.argtypes = [vp, vp, vp, i32, i32, i32, i32, C.c_float, vp, vp, vp]
        lib.dm_f32_load_clip_weight.argtypes = [vp, C.c_char_p, vp, i32, C.POINTER(i64), i32]
        lib.dm_f32_finalize_clip.argtypes = [vp]
        lib.dm_f32_clip_encode.argtypes = [vp, vp, i32, i32, vp, vp]
        lib.dm_f32_score.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    if path is None:
        _lib = lib
    return lib


def scheduler_alphas_cumprod(n: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012) -> np.ndarray:
    """Host-only: the engine's own ᾱ table (no GPU needed)."""
    lib = load_library()
    out = np.empty(n, dtype=np.float32)
    if lib.dm_scheduler_alphas_cumprod(n, beta_start, beta_end, out.ctypes.data_as(C.POINTER(C.c_float))):
        raise EngineError("dm_scheduler_alphas_cumprod failed")
    return out


def timestep_sinusoid(t: int, dim: int = 320) -> np.ndarray:
    lib = load_library()
    out = np.empty(dim, dtype=np.float32)
    if lib.dm_timestep_sinusoid(int(t), dim, out.ctypes.data_as(C.POINTER(C.c_float))):
        raise EngineError("dm_timestep_sinusoid failed")
    return out


def dift_shape(h: int, w: int, up_ft_index: int = 1) -> Tuple[int, int, int]:
    lib = load_library()
    c, oh, ow = C.c_int(), C.c_int(), C.c_int()
    if lib.dm_dift_shape(h, w, up_ft_index, C.byref(c), C.byref(oh), C.byref(ow)):
        raise EngineError("dm_dift_shape failed")
    return c.value, oh.value, ow.value


class UNetEngine:
    """One engine per GPU (the reference is one process per GPU, compute.py:215)."""

    def __init__(self, device: int = 0):
        import torch
        self._torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise EngineError("no GPU visible: the MI355X typicality engine has no CPU fallback")
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        h = C.c_void_p()
        if self.lib.dm_engine_create(self.device_index, C.byref(h)):
            raise EngineError("dm_engine_create: " + self.lib.dm_last_error(None).decode())
        self._h = h
        self.n_prompts = 0
        self.prompt_generation = 0          # bumped by every set_prompts: callers that cache slots compare it
        self._finalized = False

    # -- plumbing --------------------------------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc:
            raise EngineError(f"{what}: {self.lib.dm_last_error(self._h).decode()}")

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dm_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights ---------------------------------------------------------------------------------
    def _load(self, fn, sd, what):
        torch = self._torch
        for name, t in sd.items():
            if hasattr(t, "detach"):
                t = t.detach().to("cpu")
                t = t.to(torch.float32).numpy() if t.dtype not in (torch.float16, torch.float32) else t.numpy()
            a = np.ascontiguousarray(t)
            if a.dtype == np.float16:
                dt = 0
            elif a.dtype == np.float32:
                dt = 1
            else:
                a = a.astype(np.float32)
                dt = 1
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(fn(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), dt, shape, a.ndim), f"{what}({name})")

    def load_state_dict(self, sd: Dict[str, "np.ndarray"]):
        """sd: diffusers-named U-Net state dict (numpy or torch tensors, fp16/fp32/bf16)."""
        self._load(self.lib.dm_engine_load_weight, sd, "load_weight")
        self._check(self.lib.dm_engine_finalize(self._h), "finalize")
        self._finalized = True

    def load_vae_state_dict(self, sd: Dict[str, "np.ndarray"]):
        """sd: `AutoencoderKL.state_dict()` (`pipe.vae`, compute.py:73); only `encoder.*` and
        `quant_conv.*` are used, the decoder half is ignored.  Optional: scoring needs only the U-Net."""
        self._load(self.lib.dm_engine_load_vae_weight, sd, "load_vae_weight")
        self._check(self.lib.dm_engine_finalize_vae(self._h), "finalize_vae")
        self._vae_ready = True

    def load_clip_state_dict(self, sd: Dict[str, "np.ndarray"]):
        """sd: `CLIPTextModel.state_dict()` (`pipe.text_encoder`, compute.py:68).  Optional."""
        self._load(self.lib.dm_engine_load_clip_weight, sd, "load_clip_weight")
        self._check(self.lib.dm_engine_finalize_clip(self._h), "finalize_clip")
        self._clip_ready = True

    def clip_encode(self, input_ids, out_dtype=None):
        """`self.clip(tokens)[0]` (compute.py:51): input_ids [n, 77] (tokenizer output, padding="max_length")
        -> last_hidden_state [n, 77, 768] on the GPU (`out_dtype` fp32 default like `.float()`, or fp16)."""
        torch = self._torch
        out_dtype = out_dtype or torch.float32
        ids = torch.as_tensor(input_ids).to(self.device, torch.int32).contiguous()
        assert ids.dim() == 2 and ids.shape[1] == 77, ids.shape
        out = torch.empty(ids.shape[0], 77, 768, dtype=out_dtype, device=self.device)
        p16 = C.c_void_p(out.data_ptr()) if out_dtype == torch.float16 else None
        p32 = C.c_void_p(out.data_ptr()) if out_dtype == torch.float32 else None
        assert p16 or p32
        self._check(self.lib.dm_clip_encode(self._h, C.c_void_p(ids.data_ptr()), ids.shape[0], 77, p16, p32, self._stream()),
                    "dm_clip_encode")
        return out

    def load_vae_safetensors(self, path: str):
        """`vae/diffusion_pytorch_model.safetensors` of a diffusers pipeline directory."""
        from safetensors.numpy import load_file
        self.load_vae_state_dict(load_file(path))

    def vae_encode(self, image, noise=None, scaling_factor: float = 0.18215, out_dtype=None, return_moments=False,
                   draws_per_image: int = 1):
        """`vae.encode(image).latent_dist.sample() * scaling_factor` (compute.py:91-93) with the N(0,1) draw
        injected (`noise` [B*D,4,H/8,W/8], D = draws_per_image; None -> posterior mode).  image [B,3,H,W] in
        [-1,1].  Returns latents [B*D,4,H/8,W/8] (`out_dtype` fp16 default, or fp32; sample b*D+d belongs to
        image b) [, moments fp32 [B,8,H/8,W/8]].  The encoder runs once per image whatever D is."""
        torch = self._torch
        out_dtype = out_dtype or torch.float16
        image = image.to(self.device, torch.float16).contiguous()
        B, c, H, W = image.shape
        assert c == 3 and H >= 8 and W >= 8, image.shape
        h, w = H // 8, W // 8
        if noise is not None:
            noise = noise.to(self.device, torch.float16).contiguous()
            assert noise.shape == (B * draws_per_image, 4, h, w), noise.shape
        lat = torch.empty(B * draws_per_image, 4, h, w, dtype=out_dtype, device=self.device)
        mom = torch.empty(B, 8, h, w, dtype=torch.float32, device=self.device) if return_moments else None
        p16 = C.c_void_p(lat.data_ptr()) if out_dtype == torch.float16 else None
        p32 = C.c_void_p(lat.data_ptr()) if out_dtype == torch.float32 else None
        assert p16 or p32, "out_dtype must be torch.float16 or torch.float32"
        self._check(self.lib.dm_vae_encode(self._h, C.c_void_p(image.data_ptr()),
                                           C.c_void_p(noise.data_ptr()) if noise is not None else None, B,
                                           int(draws_per_image), H, W, float(scaling_factor), p16, p32,
                                           C.c_void_p(mom.data_ptr()) if mom is not None else None, self._stream()),
                    "dm_vae_encode")
        return (lat, mom) if return_moments else lat

    def load_safetensors(self, path: str, check_config: bool = True):
        """`unet/diffusion_pytorch_model.safetensors` of a diffusers pipeline directory (the format
        the reference's `--export-only` writes, finetuning/base.py:245-250).  The `config.json` next to it (the file
        `from_pretrained` builds the U-Net from, compute.py:65-70) is read first: an architecture other than SDv1.5's
        is rejected by key name (`unet_spec.check_unet_config`) instead of by a tensor-shape message later."""
        from safetensors.numpy import load_file
        cfg = os.path.join(os.path.dirname(os.path.abspath(path)), "config.json")
        if check_config and os.path.isfile(cfg):
            from .unet_spec import check_unet_config_file
            try:
                check_unet_config_file(cfg)
            except ValueError as e:
                raise EngineError(f"{cfg}: {e}") from None
        self.load_state_dict(load_file(path))

    def load_pipeline_dir(self, model_path: str, vae: bool = True, text_encoder: bool = True):
        """`StableDiffusionPipeline.from_pretrained(model_path)` as far as the path needs it (compute.py:65-73): the
        U-Net (config.json checked), and when present the VAE encoder and the CLIP text tower of the directory."""
        unet = os.path.join(model_path, "unet", "diffusion_pytorch_model.safetensors")
        if not os.path.isfile(unet):
            raise EngineError(f"{unet} not found (a diffusers pipeline directory with safetensors weights is expected)")
        self.load_safetensors(unet)
        v = os.path.join(model_path, "vae", "diffusion_pytorch_model.safetensors")
        if vae and os.path.isfile(v):
            self.load_vae_safetensors(v)
        c = os.path.join(model_path, "text_encoder", "model.safetensors")
        if text_encoder and os.path.isfile(c):
            from safetensors.numpy import load_file
            self.load_clip_state_dict(load_file(c))

    # -- prompts ---------------------------------------------------------------------------------
    def set_prompts(self, ctx):
        """ctx [P,77,768]: distinct prompt embeddings; precomputes cross-attention K/V for them."""
        torch = self._torch
        ctx = ctx.to(self.device, torch.float16).contiguous()
        assert ctx.dim() == 3 and ctx.shape[1] == 77 and ctx.shape[2] == 768, ctx.shape
        self._check(self.lib.dm_engine_set_prompts(self._h, C.c_void_p(ctx.data_ptr()), ctx.shape[0], self._stream()),
                    "set_prompts")
        self._ctx_keepalive = ctx
        self.n_prompts = ctx.shape[0]
        self.prompt_generation += 1

    def _slots(self, slots, batch):
        """Prompt slots of a batch -> int32 on the device.  Host-side inputs (lists, CPU tensors) are range-checked against
        the registered prompts here (a slot beyond them would read another prompt set's stale K/V rows); a tensor that
        already lives on the device is not pulled back for the check — that would be two blocking syncs in the hot path
        (`TypicalityScorer.compute_loss` passes `torch.unique`'s inverse, in range by construction) — the kernels clamp
        device-side slots to the registered range instead (memory-safe, never another engine's rows)."""
        torch = self._torch
        s = torch.as_tensor(slots)
        assert s.shape == (batch,), (s.shape, batch)
        if s.numel() and (not s.is_cuda or _CHECK_DEVICE_SLOTS):      # DM_CHECK_SLOTS=1: also range-check device tensors (two syncs)
            lo, hi = int(s.min()), int(s.max())
            if lo < 0 or hi >= self.n_prompts:
                raise EngineError(f"prompt slot {lo if lo < 0 else hi} outside the {self.n_prompts} prompts registered "
                                  "by set_prompts")
        return s.to(self.device, torch.int32).contiguous()

    def _latents(self, x, eps, latent_dtype):
        """x / eps in the element type the engine's add_noise and MSE run in: fp32 (the reference's flow, also
        the default for anything that is not fp16 already) or fp16 (both must then be fp16-valued)."""
        torch = self._torch
        if latent_dtype is None:
            latent_dtype = torch.float16 if (x.dtype == torch.float16 and eps.dtype == torch.float16) else torch.float32
        assert latent_dtype in (torch.float16, torch.float32), latent_dtype
        return (x.to(self.device, latent_dtype).contiguous(), eps.to(self.device, latent_dtype).contiguous(),
                1 if latent_dtype == torch.float32 else 0)

    # -- hot path --------------------------------------------------------------------------------
    def score(self, x, eps, t, slots, x_index=None, latent_dtype=None):
        """SD.compute_loss (compute.py:95-102) fused: returns loss [B,4,h,w] fp32 on the GPU.
        latent_dtype: torch.float32 (add_noise and the eps of the MSE in fp32, as the reference's autocast run) or
        torch.float16 (fp16 scheduler arithmetic); default: fp16 only if both x and eps already are fp16."""
        torch = self._torch
        x, eps, ld = self._latents(x, eps, latent_dtype)
        B, _, h, w = eps.shape
        t = t.to(self.device, torch.int64).contiguous()
        assert t.shape == (B,)
        s = self._slots(slots, B)
        xi = None
        if x_index is not None:
            xi = torch.as_tensor(x_index, device=self.device).to(torch.int32).contiguous()
            assert xi.shape == (B,)
        elif x.shape[0] != B:
            assert x.shape[0] == 1, "x must have 1 or B rows when x_index is not given"
            xi = torch.zeros(B, dtype=torch.int32, device=self.device)
        out = torch.empty(B, 4, h, w, dtype=torch.float32, device=self.device)
        self._check(self.lib.dm_score(self._h, C.c_void_p(x.data_ptr()),
                                      C.c_void_p(xi.data_ptr()) if xi is not None else None,
                                      C.c_void_p(eps.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(s.data_ptr()),
                                      B, x.shape[0], h, w, ld, C.c_void_p(out.data_ptr()), self._stream()), "dm_score")
        return out

    def score_conds(self, x, eps, t, n_cond: int, x_index=None, latent_dtype=None, slot_table=None):
        """D.compute_losses' inner call pattern: each of the U draws (x, eps, t) under n_cond prompts.
        Returns loss [n_cond*U,4,h,w] fp32, cond-major (row k*U+i).  Bit-identical to `score` on the tiled
        batch; the prompt-independent head of the U-Net runs once per draw.  latent_dtype as in `score`.
        slot_table [n_cond, U] int32 (optional): the registered prompt of draw i in its k-th condition — draws of images
        with different categories in one call (the reference's work list, compute.py:284-290); default: prompt k for every draw."""
        torch = self._torch
        x, eps, ld = self._latents(x, eps, latent_dtype)
        U, _, h, w = eps.shape
        t = t.to(self.device, torch.int64).contiguous()
        assert t.shape == (U,) and n_cond >= 2
        st = None
        if slot_table is not None:
            st = self._slots(torch.as_tensor(slot_table).reshape(-1), n_cond * U)
        else:
            assert n_cond <= self.n_prompts
        xi = None
        if x_index is not None:
            xi = torch.as_tensor(x_index, device=self.device).to(torch.int32).contiguous()
            assert xi.shape == (U,)
        elif x.shape[0] != U:
            assert x.shape[0] == 1
            xi = torch.zeros(U, dtype=torch.int32, device=self.device)
        out = torch.empty(n_cond * U, 4, h, w, dtype=torch.float32, device=self.device)
        self._check(self.lib.dm_score_conds_slots(self._h, C.c_void_p(x.data_ptr()),
                                                  C.c_void_p(xi.data_ptr()) if xi is not None else None,
                                                  C.c_void_p(eps.data_ptr()), C.c_void_p(t.data_ptr()),
                                                  C.c_void_p(st.data_ptr()) if st is not None else None, n_cond, U,
                                                  x.shape[0], h, w, ld, C.c_void_p(out.data_ptr()), self._stream()),
                    "dm_score_conds")
        return out

    def unet(self, sample, t, slots):
        """`unet(sample, t, ctx).sample` (compute.py:100) -> [B,4,h,w] fp16."""
        torch = self._torch
        sample = sample.to(self.device, torch.float16).contiguous()
        B, _, h, w = sample.shape
        t = torch.as_tensor(t, device=self.device).to(torch.int64).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        t = t.contiguous()
        s = self._slots(slots, B)
        out = torch.empty(B, 4, h, w, dtype=torch.float16, device=self.device)
        self._check(self.lib.dm_unet_forward(self._h, C.c_void_p(sample.data_ptr()), C.c_void_p(t.data_ptr()),
                                             C.c_void_p(s.data_ptr()), B, h, w, C.c_void_p(out.data_ptr()),
                                             self._stream()), "dm_unet_forward")
        return out

    def dift(self, noisy, t, slots, up_ft_index: int = 1, ensemble: Optional[int] = None):
        """MyUNet2DConditionModel.forward tap (dift.py:24-169).  Returns (features fp16 [B,C,h',w'],
        ensemble mean fp32 [B/ens,C,h',w'] or None)."""
        torch = self._torch
        noisy = noisy.to(self.device, torch.float16).contiguous()
        B, _, h, w = noisy.shape
        t = torch.as_tensor(t, device=self.device).to(torch.int64).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        t = t.contiguous()
        s = self._slots(slots, B)
        c, oh, ow = dift_shape(h, w, up_ft_index)
        feat = torch.empty(B, c, oh, ow, dtype=torch.float16, device=self.device)
        mean = None
        if ensemble:
            mean = torch.empty(B // ensemble, c, oh, ow, dtype=torch.float32, device=self.device)
        self._check(self.lib.dm_dift(self._h, C.c_void_p(noisy.data_ptr()), C.c_void_p(t.data_ptr()),
                                     C.c_void_p(s.data_ptr()), B, h, w, up_ft_index, C.c_void_p(feat.data_ptr()),
                                     C.c_void_p(mean.data_ptr()) if mean is not None else None,
                                     int(ensemble or 1), self._stream()), "dm_dift")
        return feat, mean

    def reduce_typicality(self, grid):
        """grid [N,n_cond,4,h,w] (fp32 or fp16, on the GPU) -> (map [h,w] fp32, scalar [1] fp32)."""
        torch = self._torch
        grid = grid.to(self.device).contiguous()
        assert grid.dim() == 5 and grid.shape[2] == 4
        assert grid.dtype in (torch.float16, torch.float32)
        N, nc, _, h, w = grid.shape
        m = torch.empty(h, w, dtype=torch.float32, device=self.device)
        sc = torch.empty(1, dtype=torch.float32, device=self.device)
        self._check(self.lib.dm_reduce_typicality(self._h, C.c_void_p(grid.data_ptr()),
                                                  1 if grid.dtype == torch.float16 else 0, N, nc, h, w,
                                                  C.c_void_p(m.data_ptr()), C.c_void_p(sc.data_ptr()), self._stream()),
                    "dm_reduce_typicality")
        return m, sc

    def reduce_typicality_batched(self, loss, n_images: int, n_draws: int, n_cond: int, cond_major: bool = False):
        """All images of a batch in one launch.  loss: [n_images, n_draws, n_cond, 4, h, w] (reference grids), or —
        cond_major — the [n_cond * n_images * n_draws, 4, h, w] rows `score_conds` returns for image-major draws.
        Returns (maps [n_images,h,w] fp32, scalars [n_images] fp32) on the GPU."""
        torch = self._torch
        loss = loss.to(self.device).contiguous()
        assert loss.dtype in (torch.float16, torch.float32)
        h, w = loss.shape[-2:]
        assert loss.shape[-3] == 4 and loss.numel() == n_images * n_draws * n_cond * 4 * h * w, loss.shape
        maps = torch.empty(n_images, h, w, dtype=torch.float32, device=self.device)
        sc = torch.empty(n_images, dtype=torch.float32, device=self.device)
        self._check(self.lib.dm_reduce_typicality_batched(self._h, C.c_void_p(loss.data_ptr()),
                                                          1 if loss.dtype == torch.float16 else 0, n_images, n_draws,
                                                          n_cond, h, w, 1 if cond_major else 0,
                                                          C.c_void_p(maps.data_ptr()), C.c_void_p(sc.data_ptr()),
                                                          self._stream()), "dm_reduce_typicality_batched")
        return maps, sc

    def typicality_image(self, grid, image_size, kx: int = 1, ky: int = 1):
        """`Cluster.load_typicality` (cluster.py:125-137) on the GPU: grid [N,n_cond,4,h,w] ->
        [H-kx+1, W-ky+1] fp32 = mean_N(pool(interp(mean_C L_null)) - pool(interp(mean_C L_c)));
        image_size = (H, W).  kx = ky = 1 is the X-ray per-pixel map (xray/compute.py:210-218)."""
        torch = self._torch
        grid = grid.to(self.device).contiguous()
        assert grid.dim() == 5 and grid.shape[2] == 4 and grid.dtype in (torch.float16, torch.float32)
        N, nc, _, h, w = grid.shape
        H, W = int(image_size[0]), int(image_size[1])
        work = torch.empty(h * w + H * W, dtype=torch.float32, device=self.device)
        out = torch.empty(H - kx + 1, W - ky + 1, dtype=torch.float32, device=self.device)
        self._check(self.lib.dm_typicality_image(self._h, C.c_void_p(grid.data_ptr()),
                                                 1 if grid.dtype == torch.float16 else 0, N, nc, h, w, H, W, kx, ky,
                                                 C.c_void_p(work.data_ptr()), C.c_void_p(out.data_ptr()), self._stream()),
                    "dm_typicality_image")
        return out

    NORM_MODES = {"signed": 1, "maxabs": 2, "positive": 3, "split": 4}

    def normalize_map(self, dm, mode: str = "signed"):
        """The consumers' normalisations of an image-space map (fp32, on the GPU, numpy's fp32 arithmetic): "signed" =
        `normalize(dm)` of cluster.py:32-47 as `load_typicality_norm` calls it (cluster.py:112-123); "maxabs" = `dm /
        np.max(np.abs(dm))` (`d_compute`, utils.py:130; utils.py:14-20); "positive" = positive_only=True; "split" =
        positive_only='split' (returns the pair)."""
        torch = self._torch
        dm = dm.to(self.device, torch.float32).contiguous()
        out = torch.empty_like(dm)
        neg = torch.empty_like(dm) if mode == "split" else None
        work = torch.empty(2, dtype=torch.float32, device=self.device)
        self._check(self.lib.dm_normalize_map(self._h, C.c_void_p(dm.data_ptr()), dm.numel(), self.NORM_MODES[mode],
                                              C.c_void_p(work.data_ptr()), C.c_void_p(out.data_ptr()),
                                              C.c_void_p(neg.data_ptr()) if neg is not None else None, self._stream()),
                    "dm_normalize_map")
        return (out, neg) if mode == "split" else out

    def patch_embed(self, feat, boxes):
        """DIFT patch descriptors (cluster.py:291-299): feat [C,h,w] or [1,C,h,w] fp32 (the ensemble mean of
        `dift`), boxes [P,4] int = (r0, r1, c0, c1) feature cells -> [P,C] fp32, window mean then L2-normalised."""
        torch = self._torch
        feat = feat.to(self.device, torch.float32).contiguous()
        if feat.dim() == 4:
            assert feat.shape[0] == 1
            feat = feat[0]
        Cc, h, w = feat.shape
        b = torch.as_tensor(boxes, device=self.device).to(torch.int32).contiguous()
        assert b.dim() == 2 and b.shape[1] == 4, b.shape
        out = torch.empty(b.shape[0], Cc, dtype=torch.float32, device=self.device)
        self._check(self.lib.dm_patch_embed(self._h, C.c_void_p(feat.data_ptr()), Cc, h, w, C.c_void_p(b.data_ptr()),
                                            b.shape[0], C.c_void_p(out.data_ptr()), self._stream()), "dm_patch_embed")
        return out

    # -- measurement -----------------------------------------------------------------------------
    def prof_enable(self, on: bool = True):
        self._check(self.lib.dm_prof_enable(self._h, 1 if on else 0), "prof_enable")

    def measure_mfma_rate(self, steps: int = 20000, zero_operands: bool = False) -> dict:
        """Measurement only: TFLOP/s and shader clock the matrix cores sustain on the igemm tile's MFMA stream alone (dm_measure_mfma_rate;
        ~10 us per 100 steps).  bench.py quotes the kernel family against it beside the nominal peak."""
        tf, ghz = C.c_double(), C.c_double()
        self._check(self.lib.dm_measure_mfma_rate(self._stream(), int(steps), 1 if zero_operands else 0, C.byref(tf), C.byref(ghz)), "measure_mfma_rate")
        return {"tflops": tf.value, "sclk_ghz": ghz.value}

    def prof_read(self) -> dict:
        a, b, c2 = C.c_double(), C.c_double(), C.c_int64()
        d, e, f = C.c_double(), C.c_double(), C.c_int64()
        self._check(self.lib.dm_prof_read(self._h, C.byref(a), C.byref(b), C.byref(c2), C.byref(d), C.byref(e),
                                          C.byref(f)), "prof_read")
        out = {"igemm_ms": a.value, "igemm_flops": b.value, "igemm_launches": c2.value,
               "attn_ms": d.value, "attn_flops": e.value, "attn_launches": f.value, "igemm_flops_folded": 0.0}
        if hasattr(self.lib, "dm_prof_read_folded"):       # (absent only from older A/B libraries loaded through DM_ENGINE_LIB)
            g = C.c_double()
            self._check(self.lib.dm_prof_read_folded(self._h, C.byref(g)), "prof_read_folded")
            out["igemm_flops_folded"] = g.value
        return out

    def reserve(self, max_batch: int = 0, h: int = 0, w: int = 0, n_cond: int = 1, max_prompts: int = 0):
        """Pre-size the workspace arena for U-Net batches of up to `max_batch` samples of h x w latents (n_cond > 1: the
        `score_conds` schedule) and the K/V cache for `max_prompts` prompts: no call within those bounds allocates
        afterwards (dm_engine_reserve)."""
        self._check(self.lib.dm_engine_reserve(self._h, int(max_batch), int(h), int(w), int(n_cond), int(max_prompts),
                                               self._stream()), "dm_engine_reserve")

    def stats(self) -> dict:
        a, b, g = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.dm_engine_stats(self._h, C.byref(a), C.byref(b), C.byref(g)), "dm_engine_stats")
        return {"device_allocs": a.value, "schedule_dry_runs": b.value, "graph_launches": g.value}

    def memory(self) -> dict:
        a, b = C.c_size_t(), C.c_size_t()
        self._check(self.lib.dm_engine_memory(self._h, C.byref(a), C.byref(b)), "memory")
        return {"weights_bytes": a.value, "arena_bytes": b.value}


class UNetEngineF32:
    """The SDv1.5 U-Net in plain fp32 on the fp32 matrix cores (C ABI: dm_f32_*) — the arithmetic of the reference's DIFT
    featuriser, which loads its pipeline without torch_dtype and runs without autocast (dift.py:191,197-199).  Same
    diffusers-named state dict and prompt-slot mechanism as `UNetEngine`; every tensor at this boundary is fp32.
    No fallback: raises without the library or a GPU."""

    def __init__(self, device: int = 0):
        import torch
        self._torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise EngineError("no GPU visible: the MI355X engine has no CPU fallback")
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        h = C.c_void_p()
        if self.lib.dm_f32_create(self.device_index, C.byref(h)):
            raise EngineError("dm_f32_create: " + self.lib.dm_f32_last_error(None).decode())
        self._h = h
        self.n_prompts = 0
        self.prompt_generation = 0

    def _check(self, rc: int, what: str):
        if rc:
            raise EngineError(f"{what}: {self.lib.dm_f32_last_error(self._h).decode()}")

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dm_f32_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, "np.ndarray"]):
        """sd: diffusers-named U-Net state dict (numpy or torch; fp32 kept as is, fp16 / bf16 widened exactly)."""
        UNetEngine._load(self, self.lib.dm_f32_load_weight, sd, "f32_load_weight")
        self._check(self.lib.dm_f32_finalize(self._h), "f32_finalize")

    def load_safetensors(self, path: str):
        from safetensors.numpy import load_file
        self.load_state_dict(load_file(path))

    def load_vae_state_dict(self, sd: Dict[str, "np.ndarray"]):
        """sd: `AutoencoderKL.state_dict()` (`pipe.vae`); only `encoder.*` and `quant_conv.*` are used.  Optional."""
        UNetEngine._load(self, self.lib.dm_f32_load_vae_weight, sd, "f32_load_vae_weight")
        self._check(self.lib.dm_f32_finalize_vae(self._h), "f32_finalize_vae")
        self._vae_ready = True

    def load_clip_state_dict(self, sd: Dict[str, "np.ndarray"]):
        """sd: `CLIPTextModel.state_dict()` (`pipe.text_encoder`); the featuriser's pipeline keeps it in fp32 (dift.py:197-199).  Optional."""
        UNetEngine._load(self, self.lib.dm_f32_load_clip_weight, sd, "f32_load_clip_weight")
        self._check(self.lib.dm_f32_finalize_clip(self._h), "f32_finalize_clip")
        self._clip_ready = True

    def clip_encode(self, input_ids, out_dtype=None):
        """`text_encoder(input_ids)[0]` in fp32 — `pipe.encode_prompt(prompt)[0]` of the reference's featuriser (dift.py:222-226):
        input_ids [n, 77] (tokenizer output, padding="max_length") -> last_hidden_state [n, 77, 768] fp32 on the GPU."""
        torch = self._torch
        if out_dtype not in (None, torch.float32):
            raise ValueError("the fp32 net's text tower returns fp32 hidden states")
        ids = torch.as_tensor(input_ids).to(self.device, torch.int32).contiguous()
        if ids.dim() != 2 or ids.shape[1] != 77:
            raise ValueError(f"input_ids must be [n, 77], got {tuple(ids.shape)}")
        out = torch.empty(ids.shape[0], 77, 768, dtype=torch.float32, device=self.device)
        self._check(self.lib.dm_f32_clip_encode(self._h, C.c_void_p(ids.data_ptr()), ids.shape[0], 77, C.c_void_p(out.data_ptr()),
                                                self._stream()), "dm_f32_clip_encode")
        return out

    def vae_encode(self, image, noise=None, scaling_factor: float = 0.18215, return_moments=False, draws_per_image: int = 1):
        """`vae.encode(image).latent_dist.sample() * scaling_factor` in fp32 (dift.py:187) with the N(0,1) draw injected (`noise`
        [B*D,4,H/8,W/8]; None -> posterior mode).  image [B,3,H,W] in [-1,1].  Returns latents [B*D,4,H/8,W/8] fp32 [, moments
        fp32 [B,8,H/8,W/8]]; the encoder runs once per image whatever D is."""
        torch = self._torch
        image = image.to(self.device, torch.float32).contiguous()
        B, c, H, W = image.shape
        assert c == 3 and H >= 8 and W >= 8, image.shape
        h, w = H // 8, W // 8
        if noise is not None:
            noise = noise.to(self.device, torch.float32).contiguous()
            assert noise.shape == (B * draws_per_image, 4, h, w), noise.shape
        lat = torch.empty(B * draws_per_image, 4, h, w, dtype=torch.float32, device=self.device)
        mom = torch.empty(B, 8, h, w, dtype=torch.float32, device=self.device) if return_moments else None
        self._check(self.lib.dm_f32_vae_encode(self._h, C.c_void_p(image.data_ptr()), C.c_void_p(noise.data_ptr()) if noise is not None else None,
                                               B, int(draws_per_image), H, W, float(scaling_factor), C.c_void_p(lat.data_ptr()),
                                               C.c_void_p(mom.data_ptr()) if mom is not None else None, self._stream()), "dm_f32_vae_encode")
        return (lat, mom) if return_moments else lat

    def set_prompts(self, ctx):
        """ctx [P,77,768] (`prompt_embeds`, dift.py:222-227), kept in fp32."""
        torch = self._torch
        ctx = ctx.to(self.device, torch.float32).contiguous()
        assert ctx.dim() == 3 and ctx.shape[1] == 77 and ctx.shape[2] == 768, ctx.shape
        self._check(self.lib.dm_f32_set_prompts(self._h, C.c_void_p(ctx.data_ptr()), ctx.shape[0], self._stream()), "f32_set_prompts")
        self._ctx_keepalive = ctx
        self.n_prompts = ctx.shape[0]
        self.prompt_generation += 1

    def _tsl(self, t, slots, B):
        torch = self._torch
        t = torch.as_tensor(t, device=self.device).to(torch.int64).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        s = torch.as_tensor(slots)
        assert s.shape == (B,) and t.shape == (B,), (s.shape, t.shape, B)
        if s.numel() and not s.is_cuda:
            lo, hi = int(s.min()), int(s.max())
            if lo < 0 or hi >= self.n_prompts:
                raise EngineError(f"prompt slot {lo if lo < 0 else hi} outside the {self.n_prompts} prompts registered by set_prompts")
        return t.contiguous(), s.to(self.device, torch.int32).contiguous()

    def unet(self, sample, t, slots):
        """`unet(sample, t, ctx).sample` in fp32 -> [B,4,h,w] fp32."""
        torch = self._torch
        sample = sample.to(self.device, torch.float32).contiguous()
        B, _, h, w = sample.shape
        t, s = self._tsl(t, slots, B)
        out = torch.empty(B, 4, h, w, dtype=torch.float32, device=self.device)
        self._check(self.lib.dm_f32_unet_forward(self._h, C.c_void_p(sample.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(s.data_ptr()),
                                                 B, h, w, C.c_void_p(out.data_ptr()), self._stream()), "dm_f32_unet_forward")
        return out

    def score(self, x, eps, t, slots, x_index=None):
        """SD.compute_loss (compute.py:95-102) with no autocast: fp32 add_noise, fp32 U-Net, fp32 squared error -> [B,4,h,w] fp32.
        The exact-arithmetic yardstick of `UNetEngine.score`."""
        torch = self._torch
        x = x.to(self.device, torch.float32).contiguous()
        eps = eps.to(self.device, torch.float32).contiguous()
        B, _, h, w = eps.shape
        t, s = self._tsl(t, slots, B)
        xi = None
        if x_index is not None:
            xi = torch.as_tensor(x_index, device=self.device).to(torch.int32).contiguous()
            assert xi.shape == (B,)
        elif x.shape[0] != B:
            assert x.shape[0] == 1, "x must have 1 or B rows when x_index is not given"
            xi = torch.zeros(B, dtype=torch.int32, device=self.device)
        out = torch.empty(B, 4, h, w, dtype=torch.float32, device=self.device)
        self._check(self.lib.dm_f32_score(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(xi.data_ptr()) if xi is not None else None,
                                          C.c_void_p(eps.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(s.data_ptr()), B, x.shape[0], h, w,
                                          C.c_void_p(out.data_ptr()), self._stream()), "dm_f32_score")
        return out

    def score_conds(self, x, eps, t, n_cond: int, x_index=None, latent_dtype=None, slot_table=None):
        """Each of the U draws (x, eps, t) under n_cond prompts -> loss [n_cond*U,4,h,w] fp32, cond-major (row k*U+i), like
        `UNetEngine.score_conds` — the same keywords, so `TypicalityScorer.compute_losses_batch` / `compute_submission` run on either
        engine: `slot_table` [n_cond, U] = the registered prompt of draw i in its k-th condition (default: prompt k for
        every draw); `latent_dtype` must be None or float32 (this net has one dtype flow: everything fp32)."""
        torch = self._torch
        if latent_dtype not in (None, torch.float32):
            raise ValueError("UNetEngineF32.score_conds: the fp32 net has no fp16 latent flow (latent_dtype must be None or torch.float32)")
        U = eps.shape[0]
        t = torch.as_tensor(t).reshape(-1)
        xi = None if x_index is None else torch.as_tensor(x_index).reshape(-1).repeat(n_cond)
        xx = x if (x.shape[0] == 1 or x_index is not None) else x.repeat(n_cond, 1, 1, 1)
        if slot_table is not None:
            slots = torch.as_tensor(slot_table).to(torch.int32).reshape(-1)
            if slots.numel() != n_cond * U:
                raise ValueError(f"slot_table must hold n_cond * U = {n_cond * U} entries, got {slots.numel()}")
        else:
            slots = torch.arange(n_cond, dtype=torch.int32).repeat_interleave(U)
        return self.score(xx, eps.repeat(n_cond, 1, 1, 1), t.repeat(n_cond), slots, x_index=xi)

    def dift(self, noisy, t, slots, up_ft_index: int = 1, ensemble: Optional[int] = None):
        """MyUNet2DConditionModel.forward tap (dift.py:24-169) in the reference's fp32.  Returns (features fp32 [B,C,h',w'],
        ensemble mean fp32 [B/ens,C,h',w'] or None)."""
        torch = self._torch
        noisy = noisy.to(self.device, torch.float32).contiguous()
        B, _, h, w = noisy.shape
        t, s = self._tsl(t, slots, B)
        c, oh, ow = dift_shape(h, w, up_ft_index)
        feat = torch.empty(B, c, oh, ow, dtype=torch.float32, device=self.device)
        mean = torch.empty(B // ensemble, c, oh, ow, dtype=torch.float32, device=self.device) if ensemble else None
        self._check(self.lib.dm_f32_dift(self._h, C.c_void_p(noisy.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(s.data_ptr()), B, h, w,
                                         up_ft_index, C.c_void_p(feat.data_ptr()), C.c_void_p(mean.data_ptr()) if mean is not None else None,
                                         int(ensemble or 1), self._stream()), "dm_f32_dift")
        return feat, mean

    def prof_enable(self, on: bool = True):
        self._check(self.lib.dm_f32_prof_enable(self._h, 1 if on else 0), "f32_prof_enable")

    def prof_read(self) -> dict:
        a, b, c2 = C.c_double(), C.c_double(), C.c_int64()
        d, e, f = C.c_double(), C.c_double(), C.c_int64()
        self._check(self.lib.dm_f32_prof_read(self._h, C.byref(a), C.byref(b), C.byref(c2), C.byref(d), C.byref(e), C.byref(f)), "f32_prof_read")
        return {"igemm_ms": a.value, "igemm_flops": b.value, "igemm_launches": c2.value,
                "attn_ms": d.value, "attn_flops": e.value, "attn_launches": f.value}

    def memory(self) -> dict:
        a, b = C.c_size_t(), C.c_size_t()
        self._check(self.lib.dm_f32_memory(self._h, C.byref(a), C.byref(b)), "f32_memory")
        return {"weights_bytes": a.value, "arena_bytes": b.value}
