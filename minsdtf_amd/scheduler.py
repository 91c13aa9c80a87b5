"""Sampler for the SD1.5 denoise loop — host-side mirror of the reference ``Scheduler``.

Same constructor / attribute / method surface as reference ``stable_diffusion/scheduler.py``
(``Scheduler(active_tcd)``, ``set_timesteps(n)``, ``timesteps``, ``signal_rates``, ``noise_rates``,
``step(latent, timestep, latent_prev)``): the deterministic branch every BASELINE configuration uses
(``scheduler.py:238-242`` and ``:272-285,308-315``) and, with ``active_tcd=True``, the TCD schedule and
stochastic step (``:136-237,286-307``; standard schedule only, no custom timestep lists).

On the GPU path the per-step arithmetic runs inside ``msd_cfg_step``; this class supplies the
schedule (float64, exactly as the reference computes it) and the per-step coefficient table the
kernel indexes with the device-side step counter.  Every step is  x' = A * x0 + B * eps + C * z  with
x0 = (x - noise[t] * eps) / signal[t];  z is the per-step Gaussian draw of the TCD sampler (C = 0 otherwise).
"""
from __future__ import annotations

import numpy as np


class Scheduler(object):
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 original_inference_steps: int = 50, active_tcd: bool = False):
        self.active_tcd = bool(active_tcd)
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
        """Non-TCD: ``linspace(0, 1000, n, endpoint=False)`` as int32, descending (scheduler.py:238-244).
        TCD: n (approximately) evenly spaced entries of the descending distillation schedule
        ``k, 2k, .. - 1`` with k = 1000 // original_inference_steps (scheduler.py:136-152,225-237)."""
        if num_inference_steps is None:
            raise ValueError("Must pass `num_inference_steps`.")
        n = int(num_inference_steps)
        if self.active_tcd:
            original_steps = self.original_inference_steps
            if original_steps > self.num_train_timesteps or n > self.num_train_timesteps:
                raise ValueError("`original_inference_steps` / `num_inference_steps` cannot exceed the training schedule")
            k = self.num_train_timesteps // original_steps
            origin = np.asarray(list(range(1, int(original_steps * 1.0) + 1))) * k - 1
            if len(origin) // n < 1 or n > original_steps:
                raise ValueError(f"`num_inference_steps`: {n} cannot be larger than `original_inference_steps`: {original_steps}")
            origin = origin[::-1].copy()
            idx = np.floor(np.linspace(0, len(origin), num=n, endpoint=False)).astype(np.int32)
            ts = origin[idx]
        else:
            ts = np.linspace(0, 1000, n, dtype=np.int32, endpoint=False)[::-1]
        self.num_inference_steps = n
        self.timesteps = ts.copy().astype(np.int32)
        self._step_index = None

    def _prev_timestep(self, index: int) -> int:
        nxt = index + 1
        if nxt < len(self.timesteps):
            return int(self.timesteps[nxt])
        # past the end the reference uses 0 for TCD and reuses `timestep` otherwise (scheduler.py:273-277)
        return 0 if self.active_tcd else int(self.timesteps[index])

    def _tcd_terms(self, prev_t: int, eta: float):
        """(signal_s, noise_s, alpha_to / alpha_s) of the TCD step towards prev_t (scheduler.py:287-299)."""
        t_s = np.floor((1.0 - eta) * prev_t).astype(np.int32)
        alpha_s = self.alphas_cumprod[t_s]
        alpha_to = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return np.sqrt(alpha_s), np.sqrt(1.0 - alpha_s), alpha_to / alpha_s

    def step(self, latent: np.ndarray, timestep: int, latent_prev: np.ndarray, eta: float = 0.3):
        """One reverse step on host arrays (float64 coefficients -> float64 result).  The TCD branch draws
        its noise from numpy's global generator, like the reference (scheduler.py:301)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self._step_index is None:
            self._step_index = int(np.nonzero(self.timesteps == timestep)[0][0])
        assert 0 <= eta <= 1.0, "gamma must be less than or equal to 1.0"
        prev_t = self._prev_timestep(self._step_index)
        last = self._step_index == self.num_inference_steps - 1
        pred_x0 = (latent_prev - self.noise_rates[timestep] * latent) / self.signal_rates[timestep]
        if self.active_tcd:
            sig_s, noi_s, ratio = self._tcd_terms(prev_t, eta)
            out = sig_s * pred_x0 + noi_s * latent
            if eta > 0.0 and not last:
                z = np.random.randn(*latent.shape).astype(np.float32)
                out = np.sqrt(ratio) * out + np.sqrt(1.0 - ratio) * z
        elif not last:
            out = self.signal_rates[prev_t] * pred_x0 + self.noise_rates[prev_t] * latent
        else:
            out = pred_x0
        self._step_index += 1
        return out

    def coefficient_table(self, timesteps=None, eta: float = 0.3) -> np.ndarray:
        """fp32 [steps][4] = {signal[t], noise[t], A, B} in execution order, the step being
        x' = A * x0 + B * eps (+ C * z, see noise_coefficients).  Non-TCD: A, B = signal / noise rate of the
        next timestep and (1, 0) on the last step; TCD: the gamma-sampling coefficients of scheduler.py:287-305."""
        ts = self.timesteps if timesteps is None else np.asarray(timesteps, dtype=np.int32)
        tab = np.zeros((len(ts), 4), dtype=np.float64)
        for i, t in enumerate(ts):
            last = i + 1 >= len(ts)
            if self.active_tcd:
                sig_s, noi_s, ratio = self._tcd_terms(0 if last else int(ts[i + 1]), eta)
                scale = 1.0 if (last or eta <= 0.0) else np.sqrt(ratio)
                a, b = scale * sig_s, scale * noi_s
            elif last:
                a, b = 1.0, 0.0
            else:
                a, b = self.signal_rates[ts[i + 1]], self.noise_rates[ts[i + 1]]
            tab[i] = (self.signal_rates[t], self.noise_rates[t], a, b)
        return tab.astype(np.float32)

    def noise_coefficients(self, timesteps=None, eta: float = 0.3) -> np.ndarray:
        """fp32 [steps]: C of x' = A x0 + B eps + C z — sqrt(1 - alpha_to/alpha_s) on every TCD step but the
        last, 0 otherwise."""
        ts = self.timesteps if timesteps is None else np.asarray(timesteps, dtype=np.int32)
        out = np.zeros(len(ts), dtype=np.float64)
        if self.active_tcd and eta > 0.0:
            for i in range(len(ts) - 1):
                out[i] = np.sqrt(1.0 - self._tcd_terms(int(ts[i + 1]), eta)[2])
        return out.astype(np.float32)

    def __len__(self):
        return self.num_train_timesteps
