"""Sampler for the SD1.5 denoise loop — host-side mirror of the reference ``Scheduler``.

Same constructor / attribute / method surface as reference ``stable_diffusion/scheduler.py``
(``Scheduler(active_tcd)``, ``set_timesteps(n)``, ``timesteps``, ``signal_rates``, ``noise_rates``,
``step(latent, timestep, latent_prev)``), restricted to the deterministic (non-TCD) branch that
every BASELINE configuration uses (``scheduler.py:238-242`` and ``:272-285,308-315``); the TCD
stochastic branch is listed as a next row in SURVEY.md §8f and raises here.

On the GPU path the per-step arithmetic runs inside ``msd_cfg_step``; this class supplies the
schedule (float64, exactly as the reference computes it) and the per-step coefficient table the
kernel indexes with the device-side step counter.
"""
from __future__ import annotations

import numpy as np


class Scheduler(object):
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 original_inference_steps: int = 50, active_tcd: bool = False):
        if active_tcd:
            raise NotImplementedError("TCD sampling (scheduler.py:136-237,286-307) is outside the accelerated path")
        self.active_tcd = False
        self.num_train_timesteps = num_train_timesteps
        self.original_inference_steps = original_inference_steps
        # scaled-linear beta schedule of latent diffusion (scheduler.py:52-55), float64
        betas = np.square(np.linspace(np.sqrt(beta_start), np.sqrt(beta_end), num_train_timesteps))
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        self.signal_rates = np.sqrt(self.alphas_cumprod)
        self.noise_rates = np.sqrt(1.0 - self.alphas_cumprod)
        self.final_alpha_cumprod = 1.0
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int32)
        self._step_index = None

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps: int):
        """``linspace(0, 1000, n, endpoint=False)`` as int32, descending (scheduler.py:238-244)."""
        if num_inference_steps is None:
            raise ValueError("Must pass `num_inference_steps`.")
        self.num_inference_steps = int(num_inference_steps)
        ts = np.linspace(0, 1000, self.num_inference_steps, dtype=np.int32, endpoint=False)
        self.timesteps = ts[::-1].copy().astype(np.int32)
        self._step_index = None

    def _prev_timestep(self, index: int) -> int:
        nxt = index + 1
        if nxt < len(self.timesteps):
            return int(self.timesteps[nxt])
        return int(self.timesteps[index])  # past the end the reference reuses `timestep` (scheduler.py:276-277)

    def step(self, latent: np.ndarray, timestep: int, latent_prev: np.ndarray, eta: float = 0.3):
        """One deterministic reverse step on host arrays (float64 coefficients -> float64 result)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self._step_index is None:
            self._step_index = int(np.nonzero(self.timesteps == timestep)[0][0])
        prev_t = self._prev_timestep(self._step_index)
        pred_x0 = (latent_prev - self.noise_rates[timestep] * latent) / self.signal_rates[timestep]
        if self._step_index != self.num_inference_steps - 1:
            out = self.signal_rates[prev_t] * pred_x0 + self.noise_rates[prev_t] * latent
        else:
            out = pred_x0
        self._step_index += 1
        return out

    def coefficient_table(self, timesteps=None) -> np.ndarray:
        """fp32 [steps][4] = {signal[t], noise[t], signal[t_prev], noise[t_prev]} in execution order.

        ``timesteps`` defaults to the full descending schedule; img2img passes the truncated list it
        actually runs.  The final row is only used through its first two entries (the kernel
        returns x0 on the last step)."""
        ts = self.timesteps if timesteps is None else np.asarray(timesteps, dtype=np.int32)
        tab = np.zeros((len(ts), 4), dtype=np.float64)
        for i, t in enumerate(ts):
            tp = int(ts[i + 1]) if i + 1 < len(ts) else int(t)
            tab[i] = (self.signal_rates[t], self.noise_rates[t], self.signal_rates[tp], self.noise_rates[tp])
        return tab.astype(np.float32)

    def __len__(self):
        return self.num_train_timesteps
