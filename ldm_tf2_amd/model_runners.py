"""DDIM sampler with classifier-free guidance on the HIP path -- host side.

Mirrors `LatentDiffusionModel` / `LatentDiffusionModelSampler`
(model_runners.py:352-509): same constructor kwargs (the YAML `ldm` section), the
same three public methods, the three models treated as opaque callables with the
contracts of SURVEY.md section 8b.  The reference draws x_T and the per-step noise
from an unseeded tf.random.normal (:466,:478); here they are explicit inputs (or
derived from `seed`), which is what makes parity checkable.

Host (this file, float64/float32 NumPy): the schedule tables -- built once, exactly
as model_runners.py:379-423 builds them.  Device (HIP kernels via ops): everything
inside the loop.  One DDIM step = one U-Net forward on [xt; xt] + one fused
CFG/DDIM-update kernel that reads its coefficients from a device table at a
device-resident index and decrements it, so a step has no host-side scalars and
the whole step can be captured once in a HIP graph and replayed N times.
"""
from __future__ import annotations

import sys

import numpy as np
import torch

from . import ops
from .autoencoder import AutoencoderKL, AutoencoderVQ


def _tf_linspace_f32(start, stop, num):
  """tf.linspace on float32 operands [TF-mem]: exact endpoints, interior
  start + delta*i with float32 delta = (stop-start)/(num-1)."""
  start, stop = np.float32(start), np.float32(stop)
  delta = np.float32((stop - start) / np.float32(num - 1))
  inner = (start + delta * np.arange(1, num - 1, dtype=np.float32)).astype(np.float32)
  return np.concatenate([[start], inner, [stop]]).astype(np.float32)


def _extract(data, t):
  """model_runners.py:28-45: cast to float32 THEN gather; shape [-1,1,1,1]."""
  return np.asarray(data).astype(np.float32)[np.asarray(t)].reshape(-1, 1, 1, 1)


def normal_latents(seed, first_index, count, shape_hwc):
  """x_T ~ N(0,1): sample i of a run is drawn from its own generator keyed by
  (seed, global sample index), so a sample's trajectory does not depend on how
  samples are spread over GPUs."""
  out = np.empty((count,) + tuple(shape_hwc), dtype=np.float32)
  for i in range(count):
    g = np.random.default_rng([int(seed), int(first_index) + i])
    out[i] = g.standard_normal(shape_hwc, dtype=np.float32)
  return out


class LatentDiffusionModel(object):

  def __init__(self, unet, autoencoder, cond_stage_model, num_steps=1000, beta_start=1e-4,
               beta_end=2e-2, v_posterior=0., scale_factor=0.18215, eta=0., num_ddim_steps=50):
    self._unet = unet
    self._autoencoder = autoencoder
    self._cond_stage_model = cond_stage_model
    self._num_steps = num_steps
    self._beta_start = beta_start
    self._beta_end = beta_end
    self._v_posterior = v_posterior
    self._scale_factor = scale_factor
    self._eta = eta
    self._num_ddim_steps = num_ddim_steps

    # model_runners.py:379-384 (linspace and square in float32, then float64)
    ls = _tf_linspace_f32(beta_start ** 0.5, beta_end ** 0.5, num_steps)
    self._betas = (ls * ls).astype(np.float32).astype(np.float64)
    self._alphas = 1. - self._betas
    self._alphas_cumprod = np.cumprod(self._alphas, axis=0)
    self._sqrt_recip_alphas_cumprod = np.sqrt(1. / self._alphas_cumprod)
    self._sqrt_recipm1_alphas_cumprod = np.sqrt(1. / self._alphas_cumprod - 1)
    # :406-409
    self._ddim_steps = np.arange(0, num_steps, num_steps // num_ddim_steps, dtype=np.int32)
    if self._num_ddim_steps < self._num_steps:
      self._ddim_steps = self._ddim_steps + 1
    if self._ddim_steps.max() >= num_steps:
      # tf.gather on CPU raises for an out-of-range index; N must divide num_steps
      raise IndexError(f"ddim step {int(self._ddim_steps.max())} out of range: num_ddim_steps="
                       f"{num_ddim_steps} must divide num_steps={num_steps}")
    alphas_cumprod = self._alphas_cumprod[self._ddim_steps]
    # :412-415 -- a_prev at index 0 is abar[0], not 1
    self._ddim_alphas_cumprod_prev = np.concatenate(
        [[self._alphas_cumprod[0]], self._alphas_cumprod[self._ddim_steps[:-1]]], axis=0)
    # :416-419
    self._ddim_sigmas = eta * np.sqrt(
        (1 - self._ddim_alphas_cumprod_prev) / (1 - alphas_cumprod) *
        (1 - alphas_cumprod / self._ddim_alphas_cumprod_prev))
    # :420-423
    self._ddim_sqrt_recip_alphas_cumprod = self._sqrt_recip_alphas_cumprod[self._ddim_steps]
    self._ddim_sqrt_recipm1_alphas_cumprod = self._sqrt_recipm1_alphas_cumprod[self._ddim_steps]

    self.device = getattr(unet, "device", torch.device("cuda:0"))
    self._tables = None

  def _device_tables(self):
    """Device copies of the schedule, made on first use: per DDIM index the row
    (c1, c2, a_prev, sigma) after the float32 cast of `_extract`, the int32 step table
    and the device-resident loop index."""
    if self._tables is None:
      coef = np.stack([self._ddim_sqrt_recip_alphas_cumprod, self._ddim_sqrt_recipm1_alphas_cumprod,
                       self._ddim_alphas_cumprod_prev, self._ddim_sigmas], axis=1).astype(np.float32)
      self._tables = (torch.from_numpy(coef).to(self.device),
                      torch.from_numpy(self._ddim_steps.copy()).to(self.device),
                      torch.zeros(1, dtype=torch.int32, device=self.device))
    return self._tables

  @property
  def _coef_dev(self):
    return self._device_tables()[0]

  @property
  def _steps_dev(self):
    return self._device_tables()[1]

  @property
  def _index_dev(self):
    return self._device_tables()[2]

  def decode_first_stage(self, latents):
    """model_runners.py:425-434: latents / scale_factor, then the autoencoder's decode
    (the division is fused into the decoder's first kernel)."""
    if isinstance(self._autoencoder, AutoencoderKL):
      outputs = self._autoencoder.decode(latents, training=False, scale_factor=self._scale_factor)
    elif isinstance(self._autoencoder, AutoencoderVQ):
      outputs = self._autoencoder.decode(latents, force_quantize=True, training=False,
                                         scale_factor=self._scale_factor)
    else:
      raise NotImplementedError("autoencoder not implemented")
    return outputs


class LatentDiffusionModelSampler(LatentDiffusionModel):

  def __init__(self, *args, use_graph=True, verbose=True, temb_table=True, **kwargs):
    super().__init__(*args, **kwargs)
    self._use_graph = use_graph
    self._use_temb_table = bool(temb_table)      # A/B: False = four temb launches + a decrement launch per step
    self._temb_tbl = None
    self._pre_dec = False
    self._verbose = verbose
    self._graph = None
    self._graph_key = None
    self.last_step_ms = None

  # ---- the steps' temb projections, once per sampler ---------------------------------
  def _temb_kwargs(self, dec_index):
    """kwargs of UNet.forward for one step: the table of every DDIM step's temb projections (built on first use: it
    depends on the step table and the weights only) and whether the step's first launch moves the loop counter.
    A U-Net without `temb_table` (any callable with the forward contract) gets neither."""
    if self._temb_tbl is None and hasattr(self._unet, "temb_table") and self._use_temb_table:
      self._temb_tbl = self._unet.temb_table(self._steps_dev).clone()     # (the U-Net's scratch may serve another sampler)
    self._pre_dec = self._temb_tbl is not None
    if self._temb_tbl is None:
      return {}
    return dict(temb_table=self._temb_tbl, pre_decrement=bool(dec_index))

  def _loop_start_index(self, n):
    """Value of the device-side counter before the first step: steps that pre-decrement start one above."""
    self._temb_kwargs(True)
    return n if self._pre_dec else n - 1

  # ---- one step on device state -----------------------------------------------------
  def _alloc_state(self, B, h, w, c):
    key = (B, h, w, c)
    if getattr(self, "_state_key", None) != key:
      dev, f32 = self.device, torch.float32
      self._xt = torch.empty(B, h, w, c, dtype=f32, device=dev)
      self._x2 = torch.empty(2 * B, h, w, c, dtype=f32, device=dev)
      self._eps = torch.empty(2 * B, h, w, c, dtype=f32, device=dev)
      self._state_key = key
      self._graph = None

  def _noise_table(self, noises, shape):
    """Per-step noise in a buffer the sampler owns (one per shape): a captured graph reads it
    at a fixed address, whatever tensor the caller passed."""
    src = torch.as_tensor(np.asarray(noises) if not isinstance(noises, torch.Tensor) else noises,
                          dtype=torch.float32)
    assert tuple(src.shape) == tuple(shape), (tuple(src.shape), tuple(shape))
    buf = getattr(self, "_noise_buf", None)
    if buf is None or tuple(buf.shape) != tuple(shape):
      buf = torch.empty(shape, dtype=torch.float32, device=self.device)
      self._noise_buf = buf
      self._graph = None
    buf.copy_(src)
    return buf

  def _step(self, guidance_scale, clip_denoised, noise_table, dec_index, pred_x0_out=None):
    """unet([xt; xt], t=steps[index]) -> CFG -> DDIM update, all on device."""
    # (paired_rows: x2 = [xt; xt], one timestep -- rows r and r + B differ only in their context, :449-452)
    # The loop counter moves at the START of a step (`dec_index`: the U-Net's first launch pre-decrements it and
    # selects the step's row of the temb table), so a loop starts from index = N and ends at 0.
    self._unet.forward(self._x2, steps=self._steps_dev, index=self._index_dev, out=self._eps, paired_rows=True,
                       **self._temb_kwargs(dec_index))
    stride = 0 if noise_table is None else noise_table[0].numel()
    ops.cfg_ddim_update(self._eps, self._xt, self._xt, self._coef_dev, self._index_dev,
                        guidance_scale, noise=noise_table, x_unet_out=self._x2,
                        dec_index=dec_index and not self._pre_dec, clip_denoised=clip_denoised,
                        noise_index_stride=stride, pred_x0_out=pred_x0_out)

  def ddim_sample(self, xt, cond, index, guidance_scale=1., clip_denoised=True,
                  return_pred_x0=False, noise=None):
    """model_runners.py:438-472 for a host-side `index`.  `noise` [B,h,w,c] replaces
    the reference's tf.random.normal draw (zeros when omitted; irrelevant at eta=0)."""
    xt = torch.as_tensor(xt, dtype=torch.float32).to(self.device).contiguous()
    B, h, w, c = xt.shape
    self._alloc_state(B, h, w, c)
    self._set_context(cond)
    self._xt.copy_(xt)
    self._x2[:B].copy_(xt)
    self._x2[B:].copy_(xt)
    self._index_dev.fill_(int(index))
    nz = None
    if noise is not None:
      nz = torch.as_tensor(noise, dtype=torch.float32).to(self.device).contiguous()[None]
    pred_x0 = torch.empty_like(self._xt) if return_pred_x0 else None
    sample = torch.empty_like(self._xt)
    self._unet.forward(self._x2, steps=self._steps_dev, index=self._index_dev, out=self._eps, paired_rows=True,
                       **self._temb_kwargs(False))
    ops.cfg_ddim_update(self._eps, self._xt, sample, self._coef_dev, self._index_dev,
                        guidance_scale, noise=nz, x_unet_out=None, dec_index=False,
                        clip_denoised=clip_denoised, noise_index_stride=0, pred_x0_out=pred_x0)
    if return_pred_x0:
      return sample, pred_x0
    return sample

  def _set_context(self, cond):
    """Always re-projects the context (never cached on the tensor's address: the allocator
    reuses a freed context's address for the next prompt).  The U-Net keeps the projections
    in its own persistent buffers, one set per context shape, so a captured graph only has
    to be rebuilt when that shape changes."""
    cond = torch.as_tensor(cond).to(self.device).contiguous()
    self._unet.set_context(cond)
    self._ctx_shape = tuple(cond.shape)

  def ddim_p_sample_loop(self, cond_model_inputs, shape, guidance_scale=5., x_T=None,
                         noises=None, seed=0, first_sample_index=0, record=None):
    """model_runners.py:474-509.  Extra inputs the reference lacks: `x_T` [B,h,w,4]
    (else N(0,1) from `seed`, keyed per global sample index), `noises` [N,B,h,w,4]
    indexed by DDIM index (only read when eta > 0), `record` (list: receives x_t
    after every step -- disables graph replay)."""
    B, h, w, c = (int(s) for s in shape)
    context = self._cond_stage_model(cond_model_inputs)                   # :475
    n = len(self._ddim_steps)
    if x_T is None:
      x_T = normal_latents(seed, first_sample_index, B, (h, w, c))
    xt = torch.as_tensor(np.asarray(x_T) if not isinstance(x_T, torch.Tensor) else x_T,
                         dtype=torch.float32).to(self.device).contiguous()
    assert tuple(xt.shape) == (B, h, w, c)
    # :480-482 concat(context[:4], context[4:]) == context
    cond_combined = context
    self._alloc_state(B, h, w, c)
    self._set_context(cond_combined)
    noise_table = None
    if self._eta != 0.:
      if noises is None:
        noises = np.stack([normal_latents(seed + 1 + i, first_sample_index, B, (h, w, c))
                           for i in range(n)])
      noise_table = self._noise_table(noises, (n, B, h, w, c))
    self._xt.copy_(xt)
    self._x2[:B].copy_(xt)
    self._x2[B:].copy_(xt)
    self._index_dev.fill_(self._loop_start_index(n))                      # :476 (index = N - 1 in the first step)

    gkey = (float(guidance_scale), noise_table is not None, self._ctx_shape)
    use_graph = self._use_graph and record is None
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    if use_graph:
      if self._graph is None or self._graph_key != gkey:
        # warm-up run on a side stream allocates every scratch buffer, then capture
        self._index_dev.fill_(n - 1)                 # (the warm-up step does not move the counter)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
          self._step(guidance_scale, False, noise_table, dec_index=False)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._xt.copy_(xt)
        self._x2[:B].copy_(xt)
        self._x2[B:].copy_(xt)
        g = torch.cuda.CUDAGraph()
        # thread-local capture mode: in a multi-GPU job the RCCL watchdog thread polls events
        # while this thread captures; only this thread's calls belong to the capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
          self._step(guidance_scale, False, noise_table, dec_index=True)
        self._graph, self._graph_key = g, gkey
        self._xt.copy_(xt)
        self._x2[:B].copy_(xt)
        self._x2[B:].copy_(xt)
        self._index_dev.fill_(self._loop_start_index(n))
      t0.record()
      for _ in range(n):                                                  # :484-502
        self._graph.replay()
      t1.record()
    else:
      t0.record()
      for _ in range(n):
        self._step(guidance_scale, False, noise_table, dec_index=True)
        if record is not None:
          record.append(self._xt.clone())
      t1.record()
    self._loop_events = (t0, t1, n)
    if self._verbose:                                                     # :503
      print(f"[INFO] Done running denoising for {self._num_ddim_steps} steps with"
            f" eta {self._eta}")
      sys.stdout.flush()
    images = self.decode_first_stage(self._xt)                            # :506
    if self._verbose:                                                     # :507
      print("[INFO] Done decoding images from the final latent variable.")
      sys.stdout.flush()
    return images

  def ddim_p_sample_loop_progressive(self, cond_model_inputs, shape, guidance_scale=5.,
                                     record_freq=5, x_T=None, noises=None, seed=0,
                                     first_sample_index=0):
    """Intended semantics of model_runners.py:511-575 (the reference version calls a method
    that does not exist, :535, and returns three values to a caller unpacking two,
    run_ldm_sampler.py:90 -- neither bug is reproduced).  Runs the same loop as
    ddim_p_sample_loop and keeps, for each record slot r < N // record_freq, the sample and
    the predicted x0 of the LAST step whose index // record_freq == r (the reference's
    insert_mask overwrites a slot on every such step, :545-553), i.e. of index r*record_freq.
    Returns (images [B,H,W,3], sample_progress [B,R,H,W,3], pred_x0_progress [B,R,H,W,3]),
    all decoded with decode_first_stage."""
    B, h, w, c = (int(s) for s in shape)
    context = self._cond_stage_model(cond_model_inputs)
    n = len(self._ddim_steps)
    num_records = n // record_freq
    if x_T is None:
      x_T = normal_latents(seed, first_sample_index, B, (h, w, c))
    xt = torch.as_tensor(np.asarray(x_T) if not isinstance(x_T, torch.Tensor) else x_T,
                         dtype=torch.float32).to(self.device).contiguous()
    self._alloc_state(B, h, w, c)
    self._set_context(context)
    noise_table = None
    if self._eta != 0.:
      if noises is None:
        noises = np.stack([normal_latents(seed + 1 + i, first_sample_index, B, (h, w, c))
                           for i in range(n)])
      noise_table = self._noise_table(noises, (n, B, h, w, c))
    self._xt.copy_(xt)
    self._x2[:B].copy_(xt)
    self._x2[B:].copy_(xt)
    self._index_dev.fill_(self._loop_start_index(n))
    sample_prog = torch.zeros(B, num_records, h, w, c, dtype=torch.float32, device=self.device)
    x0_prog = torch.zeros_like(sample_prog)
    pred_x0 = torch.empty_like(self._xt)
    for index in range(n - 1, -1, -1):
      self._step(guidance_scale, False, noise_table, dec_index=True, pred_x0_out=pred_x0)
      r = index // record_freq
      if r < num_records:                      # later (smaller) indices overwrite the slot
        sample_prog[:, r].copy_(self._xt)
        x0_prog[:, r].copy_(pred_x0)
    images = self.decode_first_stage(self._xt)
    flat = (B * num_records, h, w, c)
    # the decoder chunks large batches itself (B * N // record_freq frames: 160 at B=4, N=200)
    sp = self.decode_first_stage(sample_prog.reshape(flat))
    sp = sp.reshape(B, num_records, *sp.shape[1:])
    xp = self.decode_first_stage(x0_prog.reshape(flat))
    xp = xp.reshape(B, num_records, *xp.shape[1:])
    return images, sp, xp

  def last_loop_ms_per_step(self):
    """Device time of the last DDIM loop divided by its step count (synchronises)."""
    t0, t1, n = self._loop_events
    t1.synchronize()
    return t0.elapsed_time(t1) / n
