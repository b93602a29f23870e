"""Host-side mirror of the reference's DDIMScheduler interface
(reference src/model/diffusers/schedulers/scheduling_ddim.py:101-376).

The tables are built with the same torch ops as the reference (torch.linspace fp32 -> 1-beta ->
torch.cumprod), so ``alphas_cumprod`` is bit-identical; it is handed to the HIP library once
(dd_set_schedule).  Inside the captured DDIM loop the per-step scalars live on the device; the
torch ``step()`` below exists for callers that drive the loop themselves (eta > 0, custom models)
and for the CPU tests of the host logic.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import math

import numpy as np
import torch


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02,
                 beta_schedule: str = "linear", trained_betas=None, clip_sample: bool = False,
                 set_alpha_to_one: bool = True, steps_offset: int = 0, prediction_type: str = "epsilon", **kwargs):
        if kwargs.get("predict_epsilon") is not None:                     # deprecated spelling (:120-122)
            prediction_type = "epsilon" if kwargs["predict_epsilon"] else "sample"
        if trained_betas is not None:
            betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":                                   # :130-131
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":                            # :132-136
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "squaredcos_cap_v2":                        # :137-139, betas_for_alpha_bar :67-94 (the Glide cosine schedule)
            # beta_i = min(1 - abar((i + 1) / N) / abar(i / N), 0.999), abar(u) = cos^2((u + 0.008) / 1.008 * pi / 2): host double arithmetic per
            # entry with the libm cosine, then ONE rounding to fp32 -- the reference's order of operations, so the table is bit-identical
            abar = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
            n = num_train_timesteps
            betas = torch.tensor([min(1 - abar((i + 1) / n) / abar(i / n), 0.999) for i in range(n)])
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__.__name__}")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, trained_betas=trained_betas, clip_sample=clip_sample,
                                      set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                                      prediction_type=prediction_type)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)           # :143-144
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]   # :150
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self._acp_host = self.alphas_cumprod.clone()                      # fp32 CPU copy the HIP library is fed from

    @property
    def num_train_timesteps(self) -> int:
        return self.config.num_train_timesteps

    def __len__(self):
        return self.config.num_train_timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def hip_supported(self, eta: float = 0.0) -> Optional[str]:
        """None if the captured HIP loop implements this configuration, else the reason it does not."""
        c = self.config
        if eta != 0.0:
            return "eta != 0 (stochastic DDIM) needs per-step noise"
        if c.prediction_type != "epsilon":
            return f"prediction_type={c.prediction_type}"
        if c.clip_sample:
            return "clip_sample=True"
        if not c.set_alpha_to_one:
            return "set_alpha_to_one=False"
        if c.steps_offset != 0:
            return "steps_offset != 0"
        return None

    def set_timesteps(self, num_inference_steps: int, device=None):
        """(arange(T) * (N // T))[::-1] + steps_offset  (:215-229)."""
        self.num_inference_steps = int(num_inference_steps)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        ts = (np.arange(0, self.num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device)
        self.timesteps += self.config.steps_offset

    def _alpha_pair(self, timestep):
        prev = timestep - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, return_dict: bool = True):
        """Generic torch implementation of one reverse step (:231-353) for callers that do not use the
        captured loop.  Same formula order as the reference so fp32 results agree to round-off."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        timestep = int(timestep)
        prev_t = timestep - self.config.num_train_timesteps // self.num_inference_steps
        a_t, a_prev = self._alpha_pair(timestep)
        a_t, a_prev = a_t.to(sample.device), a_prev.to(sample.device)
        beta_t = 1 - a_t
        ptype = self.config.prediction_type
        if ptype == "epsilon":
            x0 = (sample - beta_t ** 0.5 * model_output) / a_t ** 0.5
        elif ptype == "sample":
            x0 = model_output
        elif ptype == "v_prediction":
            x0 = (a_t ** 0.5) * sample - (beta_t ** 0.5) * model_output
            model_output = (a_t ** 0.5) * model_output + (beta_t ** 0.5) * sample
        else:
            raise ValueError(f"prediction_type {ptype} must be one of epsilon, sample, v_prediction")
        if self.config.clip_sample:
            x0 = torch.clamp(x0, -1, 1)
        variance = self._get_variance(timestep, prev_t).to(sample.device)
        std = eta * variance ** 0.5
        if use_clipped_model_output:
            model_output = (sample - a_t ** 0.5 * x0) / beta_t ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * model_output
        prev = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise.")
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                             dtype=model_output.dtype)
            prev = prev + std * variance_noise
        if not return_dict:
            return (prev,)
        return dict(prev_sample=prev, pred_original_sample=x0)

    def add_noise(self, original_samples, noise, timesteps, backend=None):
        """q_sample (:355-376).  CUDA tensors go through dd_add_noise when a backend is given."""
        if backend is not None and original_samples.is_cuda:
            return backend.add_noise(original_samples, noise, timesteps)
        acp = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sa = (acp[timesteps] ** 0.5).flatten()
        sb = ((1 - acp[timesteps]) ** 0.5).flatten()
        while sa.dim() < original_samples.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original_samples + sb * noise
