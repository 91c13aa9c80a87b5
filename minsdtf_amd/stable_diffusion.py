"""``StableDiffusion`` — the reference's public pipeline API over the MI355X path.

Mirrors reference ``stable_diffusion/stable_diffusion.py`` (class ``StableDiffusionBase`` :47-568 and
``StableDiffusion`` :575-725): same constructor arguments, same ``text_to_image`` /
``image_to_image`` / ``generate_image`` signatures and defaults, same lazily-built model
properties, same return value (``uint8 (B, H, W, 3)``).  What differs is where the loop runs:

* ``generate_image`` keeps the whole denoise loop on the GPU (:class:`DenoiseEngine`): the
  cond + uncond UNet passes of one step run as ONE batch-2B forward (no op couples samples, so
  this is the same arithmetic as the reference's two ``predict_on_batch`` calls, uncond rows
  first), CFG + rescale + the sampler step are one kernel, and the step (or the whole loop when no
  per-step callback is installed) is replayed from a hipGraph.  The latent never leaves HBM.
* ``host_loop=True`` runs the reference's own control flow instead — numpy CFG / rescale /
  ``Scheduler.step`` around ``predict_on_batch`` calls (stable_diffusion.py:442-479) — which is
  what a maintainer gets by only swapping the model classes; tests use it to check that both
  routes agree.

* With ``model.shard_batch = True`` (default False = the reference's semantics: every process runs the batch it was given),
  under a ``torch.distributed`` process group ``generate_image`` treats ``batch_size`` as the GLOBAL batch and shards it over
  the ranks: one packed broadcast of rank 0's inputs, no traffic inside a step, one all-gather of the result.

Also on this path: ``image_to_image`` (VAE encoder + shortened schedule), ``inpaint`` (latent blend
inside the sampler kernel, pixel blend before the uint8 cast), the TCD sampler, and the CLIP text models behind
``encode_text`` (token ids or, with a local BPE merge list, strings).  Not here: the reference's download helpers and
its Gradio / Streamlit shells.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import _lib, engine, ops
from . import weights as wtab
from .models import (ControlNet, DiffusionModel, HintNet, ImageDecoder, ImageEncoder, TextClipEmbedding, TextEncoder, _BoundPlan,
                     _skip_hw, default_device)
from .scheduler import Scheduler

MAX_PROMPT_LENGTH = 77
# the ControlNet encoder on a side stream beside the UNet's down path (DenoiseEngine); "0": in line, in front of the UNet
CONTROLNET_OVERLAP = os.environ.get("MSD_CONTROLNET_OVERLAP", "1") != "0"


def get_timestep_embedding(timestep, batch_size, dim=320, max_period=10000):
    """Sinusoidal embedding, [cos | sin] (reference stable_diffusion.py:543-553)."""
    half = dim // 2
    freqs = np.exp(-np.log(max_period) * np.asarray(range(0, half), dtype=np.float32) / half)
    args = np.asarray([timestep], dtype=np.float32) * freqs
    embedding = np.concatenate([np.cos(args), np.sin(args)], axis=0)
    embedding = np.reshape(embedding, [1, -1])
    return np.repeat(embedding, batch_size, axis=0)


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0, epsilon=1e-05):
    """Host version of the guidance rescale (reference stable_diffusion.py:304-315)."""
    axes = tuple(range(1, len(noise_pred_text.shape)))
    std_text = np.std(noise_pred_text, axis=axes, keepdims=True)
    std_cfg = np.std(noise_cfg, axis=axes, keepdims=True) + epsilon
    noise_pred_rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * noise_pred_rescaled + (1.0 - guidance_rescale) * noise_cfg


class DenoiseEngine:
    """Device-resident denoise loop for a fixed (batch, context lengths, steps, guidance) shape."""

    def __init__(self, unet: DiffusionModel, B: int, t_cond: int, t_uncond: int, num_steps: int, guidance: float,
                 guidance_rescale: float, control_net: Optional[ControlNet] = None, hint_net: Optional[HintNet] = None,
                 use_graph: bool = True, streams: Optional[int] = None, inpaint: bool = False, tcd: bool = False):
        unet._require_weights()
        self.unet, self.B, self.num_steps = unet, B, num_steps
        self.h, self.w = unet.h, unet.w
        self.use_graph = use_graph
        dev = unet.device
        cfg = guidance > 0.0
        self.cfg = cfg
        h, w = self.h, self.w
        # Two ways to run the cond and uncond halves of a step (no op couples samples, so both are the
        # reference's two predict_on_batch calls, :442-460):
        #  * fused: ONE batch-2B forward;
        #  * dual (streams=2): two batch-B forwards on two HIP streams that fork after the previous
        #    sampler step and join before the next one.
        # The two forms measure the same within noise on MI355X at 512x512 because the small-batch
        # kernels are bounded by per-workgroup latency with idle CUs either way.  Fused is the default (one arena,
        # one kernel chain); dual stays selectable.
        if streams is None:
            streams = 1
        self.dual = bool(cfg and streams == 2)
        fuse = cfg and (t_cond == t_uncond) and not self.dual
        # passes: list of (rows in eps, NB, context length); fused = uncond rows then cond rows
        if not cfg:
            passes = [(0, B, t_cond, "cond")]
        elif fuse:
            passes = [(0, 2 * B, t_cond, "both")]
        else:
            passes = [(0, B, t_uncond, "uncond"), (B, B, t_cond, "cond")]
        self.passes = passes
        self.has_control = control_net is not None

        # ---- preparation plans: per SCHEDULE the time-embedding tables (timestep -> MLP -> every ResBlock's projection: they do
        #      not depend on the prompt, so they run when the schedule changes, not per call); per CALL contexts -> K/V^T, hint
        prep_t = engine.Plan(dev)
        prep = engine.Plan(dev)
        e_t = engine.Emitter(prep_t, unet._W)
        e_u = engine.Emitter(prep, unet._W)
        self.step_ptr = torch.zeros(2, dtype=torch.int32, device=dev)   # {step index, ticket of msd_cfg_step's in-kernel advance}
        self._sched_key = None       # schedule whose coefficient / time-embedding tables are on the device
        self._step_init: Dict[int, torch.Tensor] = {}
        self.latent = torch.zeros(B, h, w, 4, dtype=torch.float32, device=dev)
        self.coef = torch.zeros(num_steps, 4, dtype=torch.float32, device=dev)
        self.temb_in = torch.zeros(num_steps, 320, dtype=torch.float32, device=dev)
        total_u = sum(c for _, c in engine.resblock_names(False))
        table_u = prep_t.alloc(num_steps * total_u * 4)
        engine.emit_time_embedding(e_t, self.temb_in, num_steps, table_u, encoder_only=False)
        self.ctx_in: Dict[str, torch.Tensor] = {}
        ctx_kv_u, ctx_kv_c = {}, {}
        e_c = None
        table_c = total_c = None
        if self.has_control:
            control_net._require_weights()
            hint_net._require_weights()
            e_c = engine.Emitter(prep, control_net._W)
            total_c = sum(c for _, c in engine.resblock_names(True))
            table_c = prep_t.alloc(num_steps * total_c * 4)
            engine.emit_time_embedding(engine.Emitter(prep_t, control_net._W), self.temb_in, num_steps, table_c, encoder_only=True)
        for (_row0, nb, t, tag) in passes:
            st = torch.zeros(nb, t, 768, dtype=torch.float32, device=dev)
            self.ctx_in[tag] = st
            c16 = engine.Act(prep.alloc(nb * t * 768 * 2), nb, t, 1, 768)
            prep.rec(ops.cast_f32_to_bf16, x=st, out=c16.buf, n=nb * t * 768, name=f"context.{tag}.bf16")
            ctx_kv_u[tag] = engine.emit_context_kv(e_u, c16, engine.UNET_ATTN_LAYERS, prep)
            if self.has_control:
                ctx_kv_c[tag] = engine.emit_context_kv(e_c, c16, engine.ENCODER_ATTN_LAYERS, prep)
        self.hint_img = None
        hint_act = None
        if self.has_control:
            # hint computed once per image batch (stable_diffusion.py:427-441), tiled to both halves
            nb_max = max(nb for (_r, nb, _t, _g) in passes)
            self.hint_img = torch.zeros(B, 8 * h, 8 * w, 3, dtype=torch.float32, device=dev)
            e_h = engine.Emitter(prep, hint_net._W)
            hint_act = prep.act(nb_max, h, w, 320)
            engine.emit_hintnet(e_h, self.hint_img, B, 8 * h, 8 * w, hint_act, copies=nb_max // B)
        prep_t.finalize()
        prep.finalize()
        self.prep, self.prep_t = prep, prep_t

        # ---- per-step plans: one per stream (`branches`) + the sampler step (`tail`) ------------
        n = h * w * 4
        self.eps = torch.zeros((2 * B if cfg else B), n, dtype=torch.float32, device=dev)
        cols_u = engine.temb_columns(False)
        cols_c = engine.temb_columns(True)
        self.branches = []
        step = None
        # ControlNet beside the UNet's down path: its encoder reads the same latent and is independent of the UNet until the
        # 13 residuals are added after the down path (diffusion_model.py:230-234), so with one fused pass it runs as its own
        # plan (own arena, GroupNorm / split-K scratch) on a side stream: ONE fork after the sampler step, ONE join in front
        # of the zero convs.  (Two passes — a negative prompt of another length — and the two-stream mode keep it in line.)
        self.cn_plan = None
        overlap = self.has_control and CONTROLNET_OVERLAP and len(passes) == 1 and not self.dual
        for (row0, nb, t, tag) in passes:
            if step is None or self.dual:
                step = engine.Plan(dev)   # dual: each half owns its arena, the halves are live at the same time
                self.branches.append(step)
                s_u = engine.Emitter(step, unet._W, step_ptr=self.step_ptr)
                s_c = engine.Emitter(step, control_net._W, step_ptr=self.step_ptr) if self.has_control else None
            taps = None
            if self.has_control:
                # ControlNet encoder first; its 13 zero convs run inside the UNet plan, fused with the residual adds
                hint_nb = engine.Act(hint_act.buf, nb, h, w, 320)  # first nb rows of the tiled hint
                s_cf = s_c
                if overlap:
                    self.cn_plan = engine.Plan(dev)
                    s_cf = engine.Emitter(self.cn_plan, control_net._W, step_ptr=self.step_ptr)
                feats = engine.emit_controlnet_features(s_cf, self.latent, B, nb, h, w, (table_c, total_c, 0, cols_c), ctx_kv_c[tag], t,
                                                        hint_nb)
                taps = (s_c, feats)
            eps_view = _Ptr(self.eps.data_ptr() + row0 * n * 4)
            engine.emit_unet(s_u, self.latent, B, nb, h, w, (table_u, total_u, 0, cols_u), ctx_kv_u[tag], t, eps_view,
                             control_taps=taps)
        tail = engine.Plan(dev) if self.dual else step
        # inpainting (reference :469-475): the blend with the re-noised encoded image is part of the sampler kernel
        self.inpaint = None
        if inpaint:
            self.inpaint = {"init": torch.zeros(n, dtype=torch.float32, device=dev),
                            "noise": torch.zeros(B, n, dtype=torch.float32, device=dev),
                            "mask": torch.ones(n, dtype=torch.float32, device=dev)}
        ip = self.inpaint or {}
        # TCD sampler: one N(0,1) draw per step and sample, made on the host in the reference's order (prepare())
        self.step_noise = torch.zeros(num_steps, B, n, dtype=torch.float32, device=dev) if tcd else None
        self.noise_coef = torch.zeros(num_steps, dtype=torch.float32, device=dev) if tcd else None
        tail.rec(ops.cfg_step, eps=self.eps, latent=self.latent, coef=self.coef, step_ptr=self.step_ptr, batch=B, n=n,
                 num_steps=num_steps, guidance=guidance, guidance_rescale=guidance_rescale, advance=2,
                 inpaint_init=ip.get("init"), inpaint_noise=ip.get("noise"), inpaint_mask=ip.get("mask"),
                 step_noise=self.step_noise, noise_coef=self.noise_coef)
        if self.cn_plan is not None:
            self.cn_plan.finalize()   # (first: the main plan's zero convs record addresses of its feature maps)
            self._join = step.marks["controls"]
        for pl in self.branches:
            pl.finalize()
        self.tail = tail if self.dual else None
        if self.dual:
            tail.finalize()
        if self.dual or self.cn_plan is not None:
            self._side = torch.cuda.Stream(device=dev)
        self._step_graph: Optional[torch.cuda.CUDAGraph] = None
        self._loop_graph: Optional[torch.cuda.CUDAGraph] = None
        self._loop_graph_steps = 0
        self._warmed = False
        _lib.track_graph_owner(self)

    def release_graphs(self) -> None:
        """Destroy the captured step / loop graphs (re-captured on the next run_steps)."""
        self._step_graph = self._loop_graph = None
        self._loop_graph_steps = 0

    @property
    def calls(self):
        """Every launch of one sampler step, in issue order (profiling / bench helpers)."""
        out = [c for pl in self.branches for c in pl.calls]
        if self.cn_plan is not None:   # (listed in front of the UNet's calls, which is where they ran before the overlap)
            out = self.cn_plan.calls + out
        return out + (self.tail.calls if self.tail is not None else [])

    def contexts(self, unconditional_context, context) -> dict:
        """The `prepare` input for this engine's pass layout (host arrays or device tensors)."""
        if not self.cfg:
            return {"cond": context}
        if len(self.passes) == 1:
            if isinstance(context, torch.Tensor) or isinstance(unconditional_context, torch.Tensor):
                u, c = _f32_tensor(unconditional_context), _f32_tensor(context)
                dev = c.device if isinstance(context, torch.Tensor) else u.device   # ONE target: the tensor argument's device
                return {"both": torch.cat([u.to(dev), c.to(dev)], dim=0)}
            return {"both": np.concatenate([unconditional_context, context], axis=0)}
        return {"uncond": unconditional_context, "cond": context}

    def _one_step(self, main: "torch.cuda.Stream") -> None:
        """Issue one sampler step on `main` (dual: the cond half forks to the side stream and joins
        before the CFG / sampler kernel).  Works eagerly and under stream capture."""
        if self.cn_plan is not None:
            side = self._side
            side.wait_stream(main)
            self.cn_plan.run(side.cuda_stream)                          # ControlNet encoder ...
            self.branches[0].run_range(main.cuda_stream, 0, self._join)   # ... beside conv_in + the UNet's down path + mid block
            main.wait_stream(side)
            self.branches[0].run_range(main.cuda_stream, self._join)      # zero convs (+ residual adds), up path, sampler step
            return
        if not self.dual:
            self.branches[0].run(main.cuda_stream)
            return
        side = self._side
        side.wait_stream(main)
        self.branches[0].run(main.cuda_stream)
        self.branches[1].run(side.cuda_stream)
        main.wait_stream(side)
        self.tail.run(main.cuda_stream)

    # ---- graphs
    def _capture(self, fn) -> torch.cuda.CUDAGraph:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                fn(torch.cuda.current_stream())
        torch.cuda.current_stream().wait_stream(s)
        return g

    def _warm(self) -> None:
        """One eager step before the first capture (code objects load on first launch, which must
        not happen inside a stream capture); the latent and step counter are restored."""
        if self._warmed:
            return
        saved = (self.latent.clone(), self.step_ptr.clone())
        self._one_step(torch.cuda.current_stream())
        torch.cuda.synchronize()
        self.latent.copy_(saved[0])
        self.step_ptr.copy_(saved[1])
        self._warmed = True

    def run_steps(self, count: int, callback: Optional[Callable[[int], None]] = None) -> None:
        """Advance the latent by `count` sampler steps from the current device step counter."""
        if not self.use_graph:
            st = torch.cuda.current_stream()
            for i in range(count):
                self._one_step(st)
                if callback is not None:
                    st.synchronize()   # callback(iteration) fires after the step has finished, like the reference's
                    callback(i + 1)
            return
        if callback is None:
            if self._loop_graph is None or self._loop_graph_steps != count:
                def whole(stream):
                    for _ in range(count):
                        self._one_step(stream)
                self._warm()
                self._loop_graph = self._capture(whole)  # capture records, it does not execute
                self._loop_graph_steps = count
            self._loop_graph.replay()
            return
        if self._step_graph is None:
            self._warm()
            self._step_graph = self._capture(self._one_step)
        # the reference calls callback(iteration) after the step has FINISHED (stable_diffusion.py:476-479): progress
        # bars and cancellation rely on that, so wait for each replay before reporting it
        done = torch.cuda.Event()
        for i in range(count):
            self._step_graph.replay()
            done.record()
            done.synchronize()
            callback(i + 1)

    def prepare(self, contexts: Dict[str, np.ndarray], noise: np.ndarray, scheduler: Scheduler, timesteps,
                start_index: int = 0, hint_image: Optional[np.ndarray] = None, inpaint=None, step_noise=None) -> None:
        """Upload the per-call inputs and run the preparation plan.  Every array may be a host array or a (device) tensor.
        inpaint = (init_latent (1,h,w,4), noise (B,h,w,4), latent mask (h,w) or (h,w,1)) for an engine built with
        inpaint=True; step_noise = (B, num_steps, h*w*4) TCD draws made by the caller (sharded runs: the slice of the
        draws for the global batch) instead of the draws made here."""
        if self.inpaint is not None:
            init, ip_noise, mask = inpaint
            self.inpaint["init"].copy_(_f32_tensor(init).reshape(-1))
            self.inpaint["noise"].copy_(_f32_tensor(ip_noise).reshape(self.B, -1))
            m = mask.detach().cpu().numpy() if isinstance(mask, torch.Tensor) else mask
            m = np.asarray(m, dtype=np.float32)
            # preprocessed_mask keeps the reference's (width//8, height//8) resize (:301), which is only the latent's
            # (h, w) for square images; the reference's blend then fails to broadcast — fail the same way, loudly
            if m.shape[:2] != (self.h, self.w):
                raise ValueError(f"latent mask has shape {m.shape[:2]}, the latent is {(self.h, self.w)} "
                                 "(the reference's mask resize swaps width and height: non-square inpainting is unsupported)")
            m = m.reshape(self.h, self.w, 1)
            self.inpaint["mask"].copy_(torch.from_numpy(np.ascontiguousarray(np.broadcast_to(m, (self.h, self.w, 4))).reshape(-1)))
        for tag, arr in contexts.items():
            self.ctx_in[tag].copy_(_f32_tensor(arr))
        self.latent.copy_(_f32_tensor(noise))
        # the schedule's tables: uploaded when the schedule changes, not per call (pageable host -> device copies make the
        # host wait for the stream, which keeps it from queueing this job behind the previous one's last kernels)
        # (keyed by the table's VALUES: a scheduler with other betas / final alpha / eta on the same timesteps is another schedule)
        coef = scheduler.coefficient_table()
        sched_key = (tuple(int(t) for t in scheduler.timesteps), bool(getattr(scheduler, "active_tcd", False)), coef.tobytes())
        if self._sched_key != sched_key:
            self.coef.copy_(torch.from_numpy(coef))
            temb = np.concatenate([get_timestep_embedding(int(t), 1) for t in scheduler.timesteps], axis=0)
            self.temb_in.copy_(torch.from_numpy(np.ascontiguousarray(temb, dtype=np.float32)))
            self.prep_t.run(torch.cuda.current_stream().cuda_stream)
            self._sched_key = sched_key
        init = self._step_init.get(int(start_index))
        if init is None:
            init = self._step_init[int(start_index)] = torch.tensor([int(start_index), 0], dtype=torch.int32, device=self.step_ptr.device)
        self.step_ptr.copy_(init)   # {first step, ticket 0}: device -> device
        if self.step_noise is not None:
            # scheduler.py:301 draws np.random.randn(*latent.shape) once per executed step except the last
            self.noise_coef.copy_(torch.from_numpy(scheduler.noise_coefficients()))
            if step_noise is not None:
                self.step_noise.copy_(_f32_tensor(step_noise).reshape(self.B, self.num_steps, -1).transpose(0, 1))
            else:
                z = np.zeros((self.num_steps, self.B, self.h * self.w * 4), dtype=np.float32)
                for i in range(int(start_index), self.num_steps - 1):
                    z[i] = np.random.randn(self.B, self.h, self.w, 4).astype(np.float32).reshape(self.B, -1)
                self.step_noise.copy_(torch.from_numpy(z))
        if self.has_control:
            hi = _f32_tensor(hint_image)
            if hi.shape[0] != self.B:   # (one hint for the whole batch: the reference tiles it, :435)
                hi = hi.repeat(self.B // hi.shape[0], 1, 1, 1)
            self.hint_img.copy_(hi)   # the cond / uncond replicas are made on the device (emit_hintnet)
        self.prep.run(torch.cuda.current_stream().cuda_stream)


def _f32_tensor(x) -> torch.Tensor:
    """Host array or (device) tensor -> fp32 tensor for a copy_ into an engine buffer (no host round trip for tensors)."""
    if isinstance(x, torch.Tensor):
        return x.to(torch.float32)
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))


class _Ptr:
    def __init__(self, ptr):
        self.ptr = ptr


class StableDiffusionBase:
    """Base class for the stable diffusion 1.5 pipeline (reference stable_diffusion.py:47-568)."""

    def __init__(self, img_height=512, img_width=512, jit_compile=False, active_tcd=False):
        self.img_height = img_height
        self.img_width = img_width
        self._image_encoder = None
        self._text_encoder = None
        self._hint_net = None
        self._control_net = None
        self._text_clip_embedding = None
        self._diffusion_model = None
        self._image_decoder = None
        self._tokenizer = None
        self.jit_compile = jit_compile
        self.active_tcd = active_tcd
        self.scheduler = Scheduler(active_tcd=active_tcd)
        self._engines: Dict[tuple, DenoiseEngine] = {}
        self.denoise_streams = None  # None / 1: cond+uncond as one fused batch; 2: two concurrent HIP streams
        self.text_frontend = None
        self.bpe_path = None  # local copy of CLIP's bpe_simple_vocab_16e6.txt.gz (or $MSD_BPE_PATH) for string prompts
        self.unconditional_context = None  # (77, 768) embedding of the empty prompt, supplied by the caller
        # False (default, the reference's meaning of batch_size: stable_diffusion.py:384-397 tiles one prompt to the batch THIS
        # process runs): every rank of a process group runs the whole batch on its own (independent replicas).
        # True (opt in; bench.py's jobs call dist.generate_sharded directly): under an initialised torch.distributed process
        # group generate_image() treats batch_size as the GLOBAL batch and shards it over the ranks.
        self.shard_batch = False

    # ---- public entry points (reference :84-139)
    def text_to_image(self, prompt, negative_prompt=None, batch_size=1, num_steps=50, unconditional_guidance_scale=7.5,
                      embedding=None, negative_embedding=None, seed=None, control_net_image=None, guidance_rescale=0.7,
                      callback=None, **kw):
        encoded_text = self.encode_text(prompt, embedding)
        return self.generate_image(encoded_text, negative_prompt=negative_prompt, batch_size=batch_size, num_steps=num_steps,
                                   unconditional_guidance_scale=unconditional_guidance_scale, seed=seed,
                                   negative_embedding=negative_embedding, control_net_image=control_net_image,
                                   guidance_rescale=guidance_rescale, callback=callback, **kw)

    def image_to_image(self, prompt, negative_prompt=None, batch_size=1, num_steps=50, unconditional_guidance_scale=7.5,
                       embedding=None, negative_embedding=None, seed=None, control_net_image=None, reference_image=None,
                       reference_image_strength=0.8, guidance_rescale=0.7, callback=None, **kw):
        encoded_text = self.encode_text(prompt, embedding)
        return self.generate_image(encoded_text, negative_prompt=negative_prompt, batch_size=batch_size, num_steps=num_steps,
                                   unconditional_guidance_scale=unconditional_guidance_scale, seed=seed,
                                   negative_embedding=negative_embedding, control_net_image=control_net_image,
                                   reference_image=reference_image, reference_image_strength=reference_image_strength,
                                   guidance_rescale=guidance_rescale, callback=callback, **kw)

    def inpaint(self, prompt, negative_prompt=None, batch_size=1, num_steps=50, unconditional_guidance_scale=7.5,
                embedding=None, negative_embedding=None, seed=None, control_net_image=None, reference_image=None,
                reference_image_strength=0.8, inpaint_mask=None, mask_blur_strength=None, guidance_rescale=0.7,
                callback=None, **kw):
        """Reference :141-175."""
        encoded_text = self.encode_text(prompt, embedding)
        return self.generate_image(encoded_text, negative_prompt=negative_prompt, batch_size=batch_size, num_steps=num_steps,
                                   unconditional_guidance_scale=unconditional_guidance_scale, seed=seed,
                                   negative_embedding=negative_embedding, control_net_image=control_net_image,
                                   reference_image=reference_image, reference_image_strength=reference_image_strength,
                                   inpaint_mask=inpaint_mask, mask_blur_strength=mask_blur_strength,
                                   guidance_rescale=guidance_rescale, callback=callback, **kw)

    def encode_text(self, prompt, embedding_data=None):
        """Prompt -> context (77k, 768).  Accepted forms:
        * a float array: an already encoded context, returned as is;
        * an integer array of CLIP token ids, (77,) or (k, 77) (start / end / padding tokens included):
          run through the CLIP embedding + text transformer on the device;
          k chunks are concatenated along the token axis like the reference's long prompts;
        * a string: tokenizer + prompt weighting (minsdtf_amd/text.py, reference :176-215) in front of the text
          models; needs a local copy of CLIP's BPE merge list (``bpe_path`` / ``$MSD_BPE_PATH``), or a
          ``text_frontend`` (any object with ``encode(prompt, embedding_data) -> ndarray``)."""
        if isinstance(prompt, torch.Tensor):
            prompt = prompt.detach().cpu().numpy()
        if isinstance(prompt, np.ndarray):
            if np.issubdtype(prompt.dtype, np.integer):
                return self.encode_tokens(prompt)
            return np.asarray(prompt, dtype=np.float32)
        if self.text_frontend is not None:
            return np.asarray(self.text_frontend.encode(prompt, embedding_data), dtype=np.float32)
        # reference :176-215: optional textual-inversion vectors, then tokenizer + prompt weighting + the text models
        from .text import get_weighted_text_embeddings

        embedding, count = None, 0
        if embedding_data is not None and isinstance(embedding_data, str):
            embedding = self.load_embedding(embedding_data)
            if embedding is None:
                raise ValueError(f"failed to load embedding file: {embedding_data}.")
            count = embedding.shape[0]
            embedding = np.expand_dims(embedding, axis=0)
        return get_weighted_text_embeddings(self.tokenizer, self.text_clip_embedding, self.text_encoder, prompt,
                                            model_max_length=MAX_PROMPT_LENGTH, embedding=embedding,
                                            embedding_tokens_count=count, pad_token_id=49407)

    def load_embedding(self, embedding_path):
        """Textual-inversion file -> (n_vectors, 768) array or None (reference :71-82)."""
        if not os.path.exists(str(embedding_path)):
            return None
        state = torch.load(embedding_path, map_location="cpu")
        embedding = None
        for value in (state.get("string_to_param", {}) if isinstance(state, dict) else {}).values():
            if value.dtype in (torch.float32, torch.float16):
                embedding = value.detach().numpy()
        return embedding

    @property
    def tokenizer(self):
        """CLIP BPE tokenizer (reference :533-541).  The merge list is a download in the reference; here it is a
        local file: StableDiffusion.bpe_path or $MSD_BPE_PATH."""
        if self._tokenizer is None:
            path = getattr(self, "bpe_path", None) or os.environ.get("MSD_BPE_PATH")
            if not path or not os.path.exists(path):
                raise NotImplementedError(
                    "string prompts need CLIP's BPE merge list (bpe_simple_vocab_16e6.txt.gz), which cannot be downloaded here: "
                    "set StableDiffusion.bpe_path / $MSD_BPE_PATH, or pass CLIP token ids (int array (77,)), the (77k,768) "
                    "text embedding, or set StableDiffusion.text_frontend")
            from .text import SimpleTokenizer

            self._tokenizer = SimpleTokenizer(path)
        return self._tokenizer

    @tokenizer.setter
    def tokenizer(self, value):
        self._tokenizer = value

    def encode_tokens(self, tokens) -> np.ndarray:
        """CLIP token ids (k, 77) -> context (77k, 768): embedding lookup + text transformer on the device
        (reference stable_diffusion.py:488-493 for the unconditional tokens; long_prompt_weighting.py feeds
        the same two models chunk by chunk)."""
        tokens = np.asarray(tokens, dtype=np.int32).reshape(-1, MAX_PROMPT_LENGTH)
        emb = self.text_clip_embedding.predict_on_batch([tokens, self._get_pos_ids()])
        ctx = self.text_encoder.predict_on_batch(emb)
        return np.asarray(ctx, dtype=np.float32).reshape(-1, ctx.shape[-1])

    @staticmethod
    def _get_pos_ids():
        return np.asarray([list(range(MAX_PROMPT_LENGTH))], dtype=np.int32)

    def _text_models_ready(self) -> bool:
        return False

    def _get_unconditional_context(self):
        if self.unconditional_context is None:
            if self.text_frontend is not None:
                self.unconditional_context = np.asarray(self.text_frontend.encode("", None), dtype=np.float32)
            elif self._text_models_ready():
                # reference :488-493: start token + 76 end tokens through the embedding and the text encoder
                ids = np.asarray([[49406] + [49407] * (MAX_PROMPT_LENGTH - 1)], dtype=np.int32)
                self.unconditional_context = self.encode_tokens(ids)
            else:
                raise NotImplementedError(
                    "the unconditional context is CLIP's embedding of the empty prompt; pass text_encoder_ckpt=, or set "
                    "StableDiffusion.unconditional_context to a (77,768) array, or install text_frontend")
        u = np.asarray(self.unconditional_context, dtype=np.float32)
        return u[None] if u.ndim == 2 else u

    def _expand_tensor(self, text_embedding, batch_size):
        """Reference :495-503: one (T, 768) context -> (B, T, 768); a batch of contexts passes through."""
        return self._batch_of(text_embedding, batch_size, 2)

    _get_timestep_embedding = staticmethod(lambda timestep, batch_size, dim=320, max_period=10000:
                                           get_timestep_embedding(timestep, batch_size, dim, max_period))
    rescale_noise_cfg = staticmethod(rescale_noise_cfg)

    def _get_initial_diffusion_noise(self, batch_size, seed):
        """The reference draws keras.random.normal(seed) (backend RNG, :555-557); here the noise is
        numpy's PCG64 standard normal for the GLOBAL batch, so a sharded run slices the same draw."""
        rng = np.random.default_rng(seed)
        return rng.standard_normal((batch_size, self.img_height // 8, self.img_width // 8, 4)).astype(np.float32)

    @staticmethod
    def resize(image_array, new_h=None, new_w=None):
        """Bilinear resize with align-corners sampling (reference :242-275)."""
        h, w, _c = image_array.shape
        if new_h == h and new_w == w:
            return image_array
        y = np.expand_dims(np.linspace(0, h - 1, new_h), axis=-1)
        x = np.expand_dims(np.linspace(0, w - 1, new_w), axis=0)
        x0, x1 = np.clip(np.floor(x).astype(int), 0, w - 1), np.clip(np.ceil(x).astype(int), 0, w - 1)
        y0, y1 = np.clip(np.floor(y).astype(int), 0, h - 1), np.clip(np.ceil(y).astype(int), 0, h - 1)
        dx, dy = np.expand_dims(x - x0, -1), np.expand_dims(y - y0, -1)
        top = image_array[y0, x0, :] * (1.0 - dx) + image_array[y0, x1, :] * dx
        bot = image_array[y1, x0, :] * (1.0 - dx) + image_array[y1, x1, :] * dx
        return top * (1.0 - dy) + bot * dy

    @staticmethod
    def gaussian_blur(image, radius=3, h_axis=1, v_axis=2):
        """Separable binomial blur, reflected borders (reference :217-240): the 1-D filter is row
        `radius - 1` of Pascal's triangle, normalised."""
        from math import comb

        from scipy.ndimage import correlate1d

        weights = np.asarray([comb(radius - 1, k) for k in range(radius)], dtype=np.float64)
        weights = weights / np.sum(weights)
        out = correlate1d(image, weights, axis=h_axis, output=None, mode="reflect", cval=0.0, origin=0)
        return correlate1d(out, weights, axis=v_axis, output=None, mode="reflect", cval=0.0, origin=0)

    @staticmethod
    def _pixels(source, pil_mode):
        """A file path (decoded with PIL in `pil_mode`) or anything array-like -> ndarray."""
        if type(source) is str:
            from PIL import Image

            with Image.open(source) as im:
                return np.array(im.convert(pil_mode))
        return np.array(source)

    def preprocessed_mask(self, x, blur_radius=5):
        """Mask (path or HxW[xC] array, 0..255) -> (image-resolution mask (1,H,W,1) in [0,1], latent-resolution mask
        (1,·,·,1)); behaviour of reference :288-302, pinned by goldens G2e / G7: channels are averaged AFTER the bilinear
        resize, the optional binomial blur runs at image resolution, and the latent mask is a second bilinear resize of the
        blurred one to (img_width//8, img_height//8) — rows from the width, as the reference has it (equal for squares)."""
        m = self._pixels(x, "L")
        m = m[..., None] if m.ndim == 2 else m
        m = self.resize(m, self.img_height, self.img_width)
        if m.shape[-1] > 1:
            m = m.mean(axis=-1, keepdims=True)
        unit = np.asarray(m, dtype=np.float32) / 255.0
        if blur_radius is not None:
            unit = self.gaussian_blur(unit, radius=blur_radius, h_axis=0, v_axis=1)
        small = self.resize(unit, self.img_width // 8, self.img_height // 8)
        return unit[None], small[None]

    def preprocessed_image(self, x):
        """Picture (path or HxWx3 array, 0..255) -> ((1,H,W,3) in [0,1] for the pixel blend, the same in [-1,1] for the VAE
        encoder); reference :277-286."""
        px = self.resize(self._pixels(x, "RGB"), self.img_height, self.img_width)
        unit = (np.asarray(px, dtype=np.float32) / 255.0)[None, :, :, :3]
        return unit, unit * 2.0 - 1.0

    # ---- the loop (reference :317-486)
    def _batch_of(self, value, batch_size, sample_ndim):
        """One sample (rank `sample_ndim`, after dropping size-1 axes like the reference's np.squeeze, :395,:498) is
        repeated `batch_size` times; a full batch passes through."""
        value = np.squeeze(value)
        if value.ndim == sample_ndim:
            value = np.repeat(value[None], batch_size, axis=0)
        return value

    def _negative_context(self, negative_prompt, negative_embedding, batch_size):
        """(B, 77k, 768) unconditional half of the guidance pair (reference :384-392)."""
        if negative_prompt is None and negative_embedding is None:
            return np.repeat(self._get_unconditional_context(), batch_size, axis=0)
        if isinstance(negative_prompt, (np.ndarray, torch.Tensor)):
            enc = np.asarray(negative_prompt, dtype=np.float32)
        else:
            enc = self.encode_text(negative_prompt or "", negative_embedding)
        return self._batch_of(enc, batch_size, 2)

    def _hint_batch(self, control_net_image, batch_size):
        """ControlNet conditioning picture -> (B, H, W, 3) in [0,1] (reference :427-437: arrays go through the bilinear
        resize, files through PIL's)."""
        if control_net_image is None:
            return None
        if isinstance(control_net_image, np.ndarray):
            px = self.resize(control_net_image, self.img_height, self.img_width)
        else:
            from PIL import Image

            px = Image.open(control_net_image).convert("RGB").resize((self.img_width, self.img_height))
        unit = np.array(px, dtype=np.float32) / 255.0
        return np.tile(unit[None], (batch_size, 1, 1, 1))

    def generate_image(self, encoded_text, negative_prompt=None, batch_size=1, num_steps=50, unconditional_guidance_scale=7.5,
                       diffusion_noise=None, seed=None, negative_embedding=None, control_net_image=None, inpaint_mask=None,
                       mask_blur_strength=None, reference_image=None, reference_image_strength=0.8, guidance_rescale=0.0,
                       callback=None, host_loop=False, return_latent=False):
        """Reference :317-486.  With ``self.shard_batch = True`` under an initialised torch.distributed process group
        `batch_size` is the GLOBAL batch: every rank calls this with the same arguments, rank 0's inputs are broadcast, each
        rank denoises + decodes its contiguous slice and every rank returns the whole gathered batch (minsdtf_amd/dist.py).
        Default (False): the reference's meaning, this process runs all `batch_size` samples."""
        if diffusion_noise is not None and seed is not None:
            raise ValueError("`diffusion_noise` and `seed` should not both be passed to `generate_image`. `seed` is only "
                             "used to generate diffusion noise when it's not already user-specified.")
        B = batch_size
        context = self._batch_of(encoded_text, B, 2)
        unconditional_context = self._negative_context(negative_prompt, negative_embedding, B)
        noise = self._get_initial_diffusion_noise(B, seed) if diffusion_noise is None else self._batch_of(diffusion_noise, B, 3)
        self.scheduler.set_timesteps(num_steps)

        # image_to_image (reference :410-418,559-568): encode the picture, run only the last int(n*strength+0.5) steps,
        # starting from signal[t]*z0 + noise[t]*eps; a strength outside (0,1) silently means txt2img, like the reference.
        # inpainting (:406-409,469-475,484-485) needs both mask and picture: every step the latent outside the mask is
        # replaced by the encoded picture re-noised for that step, and the decoded pixels are blended once more.
        ascending = self.scheduler.timesteps[::-1]
        run_steps = num_steps
        pixel_mask = latent_mask = picture01 = encoded = None
        start_latent = noise
        if inpaint_mask is not None:
            pixel_mask, latent_mask = self.preprocessed_mask(inpaint_mask, mask_blur_strength)
        if reference_image is not None and 0.0 < reference_image_strength < 1.0:
            picture01, picture11 = self.preprocessed_image(reference_image)
            run_steps = int(num_steps * reference_image_strength + 0.5)
            t_entry = ascending[run_steps]
            ascending = ascending[:run_steps]
            encoded = self.image_encoder.predict_on_batch(picture11)
            start_latent = (self.scheduler.signal_rates[t_entry] * np.repeat(encoded, B, axis=0)
                            + self.scheduler.noise_rates[t_entry] * noise)
        inpainting = latent_mask is not None and encoded is not None
        blend_pixels = pixel_mask is not None and picture01 is not None
        start_index = num_steps - run_steps  # position of the first executed timestep in the descending schedule
        hint = self._hint_batch(control_net_image, B)
        g, phi = float(unconditional_guidance_scale), float(guidance_rescale)

        def finish(decoded, picture=picture01, mask=pixel_mask):
            """Decoder output in [-1,1] -> uint8, truncating (reference :482-486), through the pixel-space inpaint blend."""
            decoded = np.array(((decoded + 1.0) * 0.5), dtype=np.float32)
            if blend_pixels:
                decoded = picture * (1.0 - mask) + decoded * mask
            return np.clip(decoded * 255.0, 0, 255).astype("uint8")

        if host_loop:
            latent = self._host_loop(context, unconditional_context, start_latent, g, phi, hint, callback, ascending,
                                     (encoded, noise, latent_mask[0]) if inpainting else None)
            if return_latent:
                return np.asarray(latent, dtype=np.float32)
            return finish(self.image_decoder.predict_on_batch(latent))

        # ---- device loop, sharded over the process group when there is one --------------------------------
        from . import dist as mdist

        world = mdist.world_size() if getattr(self, "shard_batch", False) else 1
        tcd_global = bool(self.active_tcd and world > 1)
        per_sample, shared = {}, {}   # name -> array; insertion order = argument order of `local`
        if hint is not None:
            per_sample["hint"] = hint
        if inpainting:
            per_sample["noise"] = noise
            shared["encoded"], shared["mask"] = encoded, latent_mask[0]
        if blend_pixels and world > 1:   # rank 0's picture and mask are the ones every slice is blended with
            shared["picture"], shared["pixel_mask"] = picture01, pixel_mask
        if tcd_global:
            # the TCD sampler draws N(0,1) for the whole batch once per executed step but the last (scheduler.py:301): made
            # here for the GLOBAL batch, in that order, so that a sample's draws do not depend on the number of ranks
            zs = np.zeros((num_steps, B, noise[0].size), dtype=np.float32)
            for i in range(start_index, num_steps - 1):
                zs[i] = np.random.randn(*noise.shape).astype(np.float32).reshape(B, -1)
            per_sample["tcd"] = np.ascontiguousarray(zs.transpose(1, 0, 2))
        dev = getattr(self, "device", None) or self.diffusion_model.device
        names = list(per_sample) + list(shared)

        def local(c, u, z, *rest):
            """This rank's slice of the batch: engine for b samples -> prepare -> loop -> decode; returns a device tensor."""
            a = dict(zip(names, rest))
            b = int(z.shape[0])
            hint_b = a.get("hint")
            ip = (a["encoded"], a["noise"], a["mask"]) if inpainting else None
            tcd_z = a.get("tcd")
            eng = self._engine(b, c.shape[1], u.shape[1], num_steps, g, phi, hint_b is not None, ip is not None)
            eng.prepare(eng.contexts(u, c), z, self.scheduler, self.scheduler.timesteps, start_index, hint_b, ip, step_noise=tcd_z)
            eng.run_steps(run_steps, callback)
            if return_latent:
                return eng.latent
            if blend_pixels:   # pixel blend in fp32 before the uint8 cast
                host = [np.asarray(a[k].cpu()) for k in ("picture", "pixel_mask")] if "picture" in a else []
                return torch.from_numpy(finish(self.image_decoder.predict_on_batch(eng.latent), *host)).to(eng.latent.device)
            return self.image_decoder.decode_to_uint8(eng.latent)

        # (a one-rank group with FORCE_COLLECTIVES still takes the real exchanges: tests/test_rccl_gpu.py)
        sharded = world > 1 or (getattr(self, "shard_batch", False) and mdist.collectives_on())
        out = mdist.generate_sharded(local, context, unconditional_context, start_latent, dev,
                                     per_sample=list(per_sample.values()), shared=list(shared.values()), shard=sharded)
        flags = engine.gn_sync_flags(dev)   # (None without a cluster-GroupNorm plan on `dev`) queued behind the job, read with its D2H
        host = out.cpu().numpy()
        if flags is not None:   # a cluster GroupNorm that gave up - on ANY rank of a sharded job: raise, never return that image
            # group-wide ONLY for a job the group ran together: an independent replica (shard_batch False under a process
            # group that exists for other reasons) must not issue a collective its peers never match
            engine.check_gn_sync(flags.cpu(), device=dev, group_wide=sharded)
        return host

    def _engine(self, B, tc, tu, steps, g, phi, control, inpaint=False) -> DenoiseEngine:
        # the engine's plans (and captured hipGraphs) hold raw addresses of the packed weights: a set_weights() /
        # load_synthetic() / LoRA reload on any of the models it was built from must retire it
        wver = (self.diffusion_model.weights_version,) + ((self.control_net.weights_version, self.hint_net.weights_version)
                                                           if control else ())
        key = (B, tc, tu, steps, g, phi, control, self.denoise_streams, inpaint, self.active_tcd, wver, engine.GN_EPOCH)
        eng = self._engines.get(key)
        if eng is None:
            # one resident engine (its arenas are the big allocations): the old one goes BEFORE the new one is built, so that a
            # re-recording (another shape, new weights, a cluster-GroupNorm give-up: GN_EPOCH) never needs room for both
            if self._engines:
                import gc

                for old in self._engines.values():
                    old.release_graphs()
                old = None
                self._engines = {}
                gc.collect()
            eng = DenoiseEngine(self.diffusion_model, B, tc, tu, steps, g, phi,
                                control_net=self.control_net if control else None,
                                hint_net=self.hint_net if control else None, use_graph=self.jit_compile,
                                streams=self.denoise_streams, inpaint=inpaint, tcd=self.active_tcd)
            self._engines = {key: eng}  # one resident engine: its arenas are the big allocations
        return eng

    def _host_loop(self, context, unconditional_context, latent, g, phi, hint_image, callback, timesteps=None, inpaint=None):
        """The reference's own loop over predict_on_batch (stable_diffusion.py:442-479)."""
        if timesteps is None:
            timesteps = self.scheduler.timesteps[::-1]
        batch_size = latent.shape[0]
        hint = self.hint_net.predict_on_batch(hint_image) if hint_image is not None else None
        iteration = 0
        for _index, timestep in list(enumerate(timesteps))[::-1]:
            latent_prev = latent
            t_emb = get_timestep_embedding(timestep, batch_size)
            if g > 0.0:
                if hint is not None:
                    uc = self.control_net.predict_on_batch([latent, t_emb, unconditional_context, hint])
                    u = self.diffusion_model.predict_on_batch([latent, t_emb, unconditional_context] + list(uc))
                    cc = self.control_net.predict_on_batch([latent, t_emb, context, hint])
                    c = self.diffusion_model.predict_on_batch([latent, t_emb, context] + list(cc))
                else:
                    u = self.diffusion_model.predict_on_batch([latent, t_emb, unconditional_context])
                    c = self.diffusion_model.predict_on_batch([latent, t_emb, context])
                latent = u + g * (c - u)
                if phi > 0.0:
                    latent = rescale_noise_cfg(latent, c, guidance_rescale=phi)
            else:
                if hint is not None:
                    cc = self.control_net.predict_on_batch([latent, t_emb, context, hint])
                    latent = self.diffusion_model.predict_on_batch([latent, t_emb, context] + list(cc))
                else:
                    latent = self.diffusion_model.predict_on_batch([latent, t_emb, context])
            latent = self.scheduler.step(latent, timestep, latent_prev)
            if inpaint is not None:   # reference :469-475
                init_latent, noise, latent_mask = inpaint
                origin = (self.scheduler.signal_rates[timestep] * np.repeat(init_latent, batch_size, axis=0)
                          + self.scheduler.noise_rates[timestep] * noise)
                latent = origin * (1.0 - latent_mask[None]) + latent * latent_mask[None]
            iteration += 1
            if callback is not None:
                callback(iteration)
        return latent


class StableDiffusion(StableDiffusionBase):
    """Reference ``StableDiffusion`` (stable_diffusion.py:575-725) with HIP-backed models; `batch_size` means what it means in the
    reference (the samples THIS process runs) unless ``shard_batch`` is set to True, which makes it the global batch of the
    initialised torch.distributed process group ("More than one GPU")."""

    def __init__(self, img_height=512, img_width=512, jit_compile=False, clip_skip=-1, unet_ckpt=None, text_encoder_ckpt=None,
                 vae_ckpt=None, lora_path=None, controlnet_path=None, active_tcd=False, device=None):
        super().__init__(img_height, img_width, jit_compile, active_tcd)
        self.clip_skip = clip_skip
        self.unet_ckpt = unet_ckpt
        self.text_encoder_ckpt = text_encoder_ckpt
        self.vae_ckpt = vae_ckpt
        self.controlnet_path = controlnet_path
        self.lora_path = None
        self.text_encoder_lora_dict = None
        self.unet_lora_dict = None
        if lora_path is not None and os.path.exists(str(lora_path)):   # reference :641-643
            self.text_encoder_lora_dict, self.unet_lora_dict = wtab.load_weights_from_lora(lora_path)
            self.lora_path = lora_path
        self.device = device if device is not None else default_device()

    @property
    def diffusion_model(self):
        if self._diffusion_model is None:
            self._diffusion_model = DiffusionModel(self.img_height, self.img_width, ckpt_path=self.unet_ckpt,
                                                   apply_control_net=self.controlnet_path is not None,
                                                   lora_dict=self.unet_lora_dict, device=self.device)
            if self.jit_compile:
                self._diffusion_model.compile(jit_compile=True)
        return self._diffusion_model

    @property
    def image_decoder(self):
        if self._image_decoder is None:
            self._image_decoder = ImageDecoder(ckpt_path=self.vae_ckpt, device=self.device)
            if self.jit_compile:
                self._image_decoder.compile(jit_compile=True)
        return self._image_decoder

    @property
    def text_clip_embedding(self):
        """Reference :686-691."""
        if self._text_clip_embedding is None:
            self._text_clip_embedding = TextClipEmbedding(MAX_PROMPT_LENGTH, ckpt_path=self.text_encoder_ckpt, device=self.device)
            if self.jit_compile:
                self._text_clip_embedding.compile(jit_compile=True)
        return self._text_clip_embedding

    @property
    def text_encoder(self):
        """Reference :672-683."""
        if self._text_encoder is None:
            self._text_encoder = TextEncoder(MAX_PROMPT_LENGTH, clip_skip=self.clip_skip, ckpt_path=self.text_encoder_ckpt,
                                             lora_dict=self.text_encoder_lora_dict, device=self.device)
            if self.jit_compile:
                self._text_encoder.compile(jit_compile=True)
        return self._text_encoder

    def _text_models_ready(self) -> bool:
        if self._text_encoder is not None and self._text_clip_embedding is not None:
            return self._text_encoder._W is not None and self._text_clip_embedding._W is not None
        return self.text_encoder_ckpt is not None and os.path.exists(str(self.text_encoder_ckpt))

    @property
    def image_encoder(self):
        if self._image_encoder is None:
            self._image_encoder = ImageEncoder(ckpt_path=self.vae_ckpt, device=self.device)
            if self.jit_compile:
                self._image_encoder.compile(jit_compile=True)
        return self._image_encoder

    @property
    def control_net(self):
        if self._control_net is None:
            self._control_net = ControlNet(self.img_height, self.img_width, controlnet_path=self.controlnet_path, device=self.device)
            if self.jit_compile:
                self._control_net.compile(jit_compile=True)
        return self._control_net

    @property
    def hint_net(self):
        if self._hint_net is None:
            self._hint_net = HintNet(self.img_height, self.img_width, controlnet_path=self.controlnet_path, device=self.device)
            if self.jit_compile:
                self._hint_net.compile(jit_compile=True)
        return self._hint_net
