"""Host-side mirror of the reference's DIFT featuriser (diffmining/typicality/dift.py:173-232) over
the HIP engine.

`SDFeaturizer.forward(latents, prompt_embeds, t=261, up_ft_index=1, ensemble_size=8)` keeps the
reference's argument meaning and its output `[1, C, H/16, W/16]` (for up_ft_index=1), with two
stated differences: the input is the scaled VAE latent (the VAE encode of dift.py:187 is outside
the path; the reference draws `ensemble_size` posterior samples of the same image there), and the
prompt arrives as CLIP hidden states `[1,77,768]` rather than as a string (text tower is outside).
"""
from __future__ import annotations

from typing import Optional

import torch

from .engine import UNetEngine


def scheduler_alphas_cumprod(n: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class SDFeaturizer:
    def __init__(self, engine: UNetEngine):
        self.engine = engine
        self.device = engine.device
        self.acp = scheduler_alphas_cumprod().to(self.device)

    def add_noise(self, latents, noise, t):
        """DDIMScheduler.add_noise in the latents' dtype (dift.py:190; fp32 in the reference)."""
        a = self.acp.to(latents.dtype)[t]
        return (a ** 0.5) * latents + ((1 - a) ** 0.5) * noise

    @torch.no_grad()
    def forward(self, latents, prompt_embeds, t: int = 261, up_ft_index: int = 1, ensemble_size: int = 8,
                noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None):
        """latents [1,4,h,w] (repeated `ensemble_size` times, dift.py:220) or [ensemble_size,4,h,w]
        (independent posterior draws, dift.py:187); prompt_embeds [1,77,768].
        Returns the ensemble-mean feature [1, C, h', w'] fp32 (dift.py:231)."""
        lat = latents.to(self.device, torch.float32)
        if lat.shape[0] == 1:
            lat = lat.repeat(ensemble_size, 1, 1, 1)
        assert lat.shape[0] == ensemble_size
        if noise is None:
            noise = torch.randn(lat.shape, generator=generator, dtype=torch.float32,
                                device=generator.device if generator is not None else "cpu")
        noisy = self.add_noise(lat, noise.to(self.device, torch.float32), int(t))
        self.engine.set_prompts(prompt_embeds.reshape(1, 77, -1))
        slots = torch.zeros(ensemble_size, dtype=torch.int32, device=self.device)
        _, mean = self.engine.dift(noisy.to(torch.float16), torch.tensor(int(t)), slots, up_ft_index, ensemble_size)
        return mean

    __call__ = forward
