"""CPU ORACLE for the latent-diffusion sampling path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch-CPU / NumPy, float32 or
float64) of the arithmetic of chao-ji/ldm_tf2's sampling path.  It is the
checker for the HIP path; nothing under `ldm_tf2_amd/` imports it, and only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may.

PARITY STATUS: **parity unpinned** for the floating-point model arithmetic.
The reference is pure Python on TensorFlow 2.13/Keras, neither of which is
installed here (ModuleNotFoundError; no network), and the reference ships no
tests or numeric fixtures.  What IS pinned by the reference (tests/golden):
the integer DDIM step tables (convert_ckpt_pytorch_to_tf2.py:402 -> 981), the
BERT token ids of the default/empty prompt (:384-392), the parameter totals
(README.md:33) and the build-probe shapes (:393-411).  Keras/TF op semantics
are restated from memory of TF 2.13 ([TF-mem]) and cross-checked against
torch.nn.functional in tests/test_oracle.py.

Every function cites the reference file:line it follows.  Layouts are the
reference's: activations NHWC, conv kernels HWIO, dense kernels [in, out],
attention projections [D, H, S] (split) / [H, S, D] (merge).
Weights come in as a flat dict name -> ndarray, names as in
`ldm_tf2_amd/weights.py` (the reference's variable structure).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------


def _t(a, dtype):
  if isinstance(a, torch.Tensor):
    return a.to(dtype)
  return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


class W:
  """Prefix view over the flat weight dict, converting lazily to torch."""

  def __init__(self, weights, dtype, prefix=""):
    self.w, self.dtype, self.p = weights, dtype, prefix

  def __call__(self, name):
    return _t(self.w[self.p + name], self.dtype)

  def has(self, name):
    return (self.p + name) in self.w

  def has_prefix(self, name):
    q = self.p + name
    return any(k.startswith(q) for k in self.w)

  def sub(self, name):
    return W(self.w, self.dtype, self.p + name + "/")


# ----------------------------------------------------------------------------
# L1 ops (third-party in the reference: tf.keras.layers.* / tf.nn.*)  [TF-mem]
# ----------------------------------------------------------------------------

def conv2d(x, kernel, bias, stride=1, pad=((1, 1), (1, 1))):
  """Keras Conv2D on NHWC with HWIO kernel.  `SAME` 3x3 stride 1 == symmetric
  pad 1; the U-Net downsample pads explicitly then runs VALID (unet.py:26-27)."""
  xn = x.permute(0, 3, 1, 2)
  (pt, pb), (pl, pr) = pad
  xn = F.pad(xn, (pl, pr, pt, pb))
  y = F.conv2d(xn, kernel.permute(3, 2, 0, 1), bias, stride=stride)
  return y.permute(0, 2, 3, 1).contiguous()


def dense(x, kernel, bias=None):
  """Keras Dense: contracts the last axis with kernel [in, out]."""
  y = x @ kernel
  return y if bias is None else y + bias


def group_norm(x, gamma, beta, groups=32, eps=1e-5):
  """Keras GroupNormalization on NHWC: reshape to [B,H,W,G,C/G], biased variance
  over (H,W,C/G), (x-mu)*rsqrt(var+eps)*gamma+beta, contiguous channel groups."""
  b, h, w, c = x.shape
  xg = x.reshape(b, h * w, groups, c // groups)
  mu = xg.mean(dim=(1, 3), keepdim=True)
  var = ((xg - mu) ** 2).mean(dim=(1, 3), keepdim=True)
  y = (xg - mu) * torch.rsqrt(var + eps)
  return y.reshape(b, h, w, c) * gamma + beta


def layer_norm(x, gamma, beta, eps=1e-5):
  """Keras LayerNormalization over the last axis, biased variance."""
  mu = x.mean(dim=-1, keepdim=True)
  var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
  return (x - mu) * torch.rsqrt(var + eps) * gamma + beta


def silu(x):
  return x * torch.sigmoid(x)


def gelu(x):
  """tf.nn.gelu default approximate=False: exact erf form."""
  return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def upsample_nearest2x(x):
  """tf.raw_ops.ResizeNearestNeighbor(2x, align_corners=False):
  out[i, j] = in[i // 2, j // 2]   (unet.py:44, autoencoder.py:152)."""
  return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


# ----------------------------------------------------------------------------
# transformer.py : Projection / text encoder
# ----------------------------------------------------------------------------

def projection_split(x, kernel):
  """transformer.py:70  einsum('NTD,DHS->NTHS')."""
  return torch.einsum("ntd,dhs->nths", x, kernel)


def projection_merge(x, kernel, bias):
  """transformer.py:68,71-72  einsum('NTHS,HSD->NTD') + bias[D]."""
  return torch.einsum("nths,hsd->ntd", x, kernel) + bias


def multihead_attention(query, context, w, scale_dim):
  """unet.py:267-292 (CrossAttention.call) == transformer.py:97-120
  (Attention.call): q/k/v split projections, logits = einsum(NQHS,NCHS->NHQC)
  THEN * S**-0.5 (unet.py:280-281), softmax over C, einsum(NHQC,NCHS->NQHS),
  merge projection with bias.  attention_mask is always None on this path."""
  q = projection_split(query, w("query/kernel"))
  k = projection_split(context, w("key/kernel"))
  v = projection_split(context, w("value/kernel"))
  logits = torch.einsum("nqhs,nchs->nhqc", q, k)
  logits = logits * (q.shape[-1] ** -0.5)   # size_per_head == the kernel's S axis
  p = torch.softmax(logits, dim=3)
  o = torch.einsum("nhqc,nchs->nqhs", p, v)
  return projection_merge(o, w("output/kernel"), w("output/bias"))


def text_encoder(token_ids, weights, dtype=torch.float32, num_heads=8,
                 size_per_head=64):
  """transformer.py:254-272 -> Encoder (:211-215) -> EncoderLayer (:173-182):
  tok_emb[ids] + pos_emb[0..T-1]; per layer x += MHA(LN(x)); x += FFN(LN(x))
  with exact-erf gelu (:169); final LN; eps 1e-5; no mask (:255)."""
  w = W(weights, dtype)
  ids = torch.as_tensor(np.asarray(token_ids), dtype=torch.long)
  t = ids.shape[1]
  x = w("embedding")[ids] + w("positional_embedding")[:t][None]
  i = 0
  while w.has(f"encoder/layers/{i}/mha/query/kernel"):
    lw = w.sub(f"encoder/layers/{i}")
    q = layer_norm(x, lw("layernorm_mha/gamma"), lw("layernorm_mha/beta"))
    x = multihead_attention(q, q, lw.sub("mha"), size_per_head) + x
    f = layer_norm(x, lw("layernorm_ffn/gamma"), lw("layernorm_ffn/beta"))
    f = gelu(dense(f, lw("ffn/filter/kernel"), lw("ffn/filter/bias")))
    x = dense(f, lw("ffn/output/kernel"), lw("ffn/output/bias")) + x
    i += 1
  return layer_norm(x, w("encoder/layernorm/gamma"), w("encoder/layernorm/beta"))


# ----------------------------------------------------------------------------
# unet.py
# ----------------------------------------------------------------------------

def get_time_embedding(time, channels, dtype=torch.float32, max_time=10000):
  """unet.py:401-422: freqs = exp(-ln(max_time) * arange(half)/half) in f32,
  emb = [cos(t*f), sin(t*f)]  -- cos first."""
  half = channels // 2
  t32 = torch.float32
  freqs = torch.exp(-torch.log(torch.tensor(float(max_time), dtype=t32)) *
                    torch.arange(0, half, dtype=t32) / half)
  args = torch.as_tensor(np.asarray(time)).to(t32)[:, None] * freqs[None]
  emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
  if channels % 2:
    emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
  return emb.to(dtype)


def unet_residual_block(x, temb, w):
  """unet.py:382-398: GN(1e-5)->SiLU->conv ; += Dense(SiLU(temb))[:,None,None]
  (additive only) ; GN->SiLU->(dropout off)->conv ; 1x1 Dense shortcut iff
  C_in != C_out ; + input."""
  h = silu(group_norm(x, w("group_norm_1/gamma"), w("group_norm_1/beta"), eps=1e-5))
  h = conv2d(h, w("conv2d_1/kernel"), w("conv2d_1/bias"))
  te = dense(silu(temb), w("dense/kernel"), w("dense/bias"))
  h = h + te[:, None, None, :]
  h = silu(group_norm(h, w("group_norm_2/gamma"), w("group_norm_2/beta"), eps=1e-5))
  h = conv2d(h, w("conv2d_2/kernel"), w("conv2d_2/bias"))
  if x.shape[-1] != h.shape[-1]:
    x = dense(x, w("shortcut/kernel"), w("shortcut/bias"))
  return h + x


def unet_transformer_block(x, context, w, size_per_head):
  """unet.py:308-314 BasicTransformerBlock + :323-325 GEGLU (value half first,
  gate second, exact gelu) + :335-338 FeedForward; LN eps 1e-5 (:304-306)."""
  h = layer_norm(x, w("layernorm1/gamma"), w("layernorm1/beta"))
  x = multihead_attention(h, h, w.sub("att_layer1"), size_per_head) + x
  h = layer_norm(x, w("layernorm2/gamma"), w("layernorm2/beta"))
  x = multihead_attention(h, context, w.sub("att_layer2"), size_per_head) + x
  h = layer_norm(x, w("layernorm3/gamma"), w("layernorm3/beta"))
  g = dense(h, w("ffn/geglu/kernel"), w("ffn/geglu/bias"))
  a, gate = torch.chunk(g, 2, dim=-1)
  h = dense(a * gelu(gate), w("ffn/dense/kernel"), w("ffn/dense/bias"))
  return h + x


def unet_spatial_transformer(x, context, w, num_heads):
  """unet.py:356-365: GN(eps 1e-6, no activation) -> Dense -> [B,hw,C] ->
  block -> reshape -> Dense -> + input."""
  b, hh, ww, c = x.shape
  h = group_norm(x, w("groupnorm/gamma"), w("groupnorm/beta"), eps=1e-6)
  h = dense(h, w("dense1/kernel"), w("dense1/bias")).reshape(b, hh * ww, c)
  h = unet_transformer_block(h, context, w.sub("block"), c // num_heads)
  h = dense(h.reshape(b, hh, ww, c), w("dense2/kernel"), w("dense2/bias"))
  return h + x


def unet_forward(x, time, context, weights, dtype=torch.float32, num_heads=8,
                 taps=None):
  """unet.py:118-138 (+ Input/Middle/OutputBlock :141-245, Downsample :24-30,
  Upsample :42-48).  x [R,h,w,4], time int [R], context [R,77,D] -> [R,h,w,4].
  The block structure is recovered from which variables exist in `weights`."""
  w = W(weights, dtype)
  x = _t(x, dtype)
  context = _t(context, dtype)
  mc = w("conv_in/bias").shape[0]
  h = conv2d(x, w("conv_in/kernel"), w("conv_in/bias"))
  temb = get_time_embedding(time, mc, dtype)
  temb = dense(silu(dense(temb, w("time_dense1/kernel"), w("time_dense1/bias"))),
               w("time_dense2/kernel"), w("time_dense2/bias"))
  hiddens = [h]
  i = 0
  while w.has_prefix(f"input_blocks/{i}/"):
    bw = w.sub(f"input_blocks/{i}")
    if bw.has("downsample/conv/kernel"):
      # unet.py:26-27 zero-pad (1,1),(1,1) then 3x3 stride-2 VALID
      h = conv2d(h, bw("downsample/conv/kernel"), bw("downsample/conv/bias"),
                 stride=2, pad=((1, 1), (1, 1)))
    else:
      h = unet_residual_block(h, temb, bw.sub("residual"))
      if bw.has_prefix("spatial_transformer/"):
        h = unet_spatial_transformer(h, context, bw.sub("spatial_transformer"), num_heads)
    if taps is not None:
      taps[f"input_blocks/{i}"] = h
    hiddens.append(h)
    i += 1
  mw = w.sub("middle_block")
  h = unet_residual_block(h, temb, mw.sub("residual1"))
  h = unet_spatial_transformer(h, context, mw.sub("spatial_transformer"), num_heads)
  h = unet_residual_block(h, temb, mw.sub("residual2"))
  if taps is not None:
    taps["middle_block"] = h
  i = 0
  while w.has_prefix(f"output_blocks/{i}/"):
    bw = w.sub(f"output_blocks/{i}")
    h = torch.cat([h, hiddens.pop()], dim=-1)          # unet.py:135
    h = unet_residual_block(h, temb, bw.sub("residual"))
    if bw.has_prefix("spatial_transformer/"):
      h = unet_spatial_transformer(h, context, bw.sub("spatial_transformer"), num_heads)
    if bw.has("upsample/conv/kernel"):
      h = conv2d(upsample_nearest2x(h), bw("upsample/conv/kernel"), bw("upsample/conv/bias"))
    if taps is not None:
      taps[f"output_blocks/{i}"] = h
    i += 1
  h = silu(group_norm(h, w("groupnorm/gamma"), w("groupnorm/beta"), eps=1e-5))
  return conv2d(h, w("conv_out/kernel"), w("conv_out/bias"))


# ----------------------------------------------------------------------------
# autoencoder.py (decode side) + quantize.py
# ----------------------------------------------------------------------------

AE_GN_EPS = 1e-6  # autoencoder.py:11


def ae_residual_block(x, w):
  """autoencoder.py:42-58 with time=None: GN->swish->conv->GN->swish->conv,
  Dense 1x1 shortcut iff C_in != channels (:53)."""
  h = conv2d(silu(group_norm(x, w("group_norm1/gamma"), w("group_norm1/beta"), eps=AE_GN_EPS)),
             w("conv1/kernel"), w("conv1/bias"))
  h = conv2d(silu(group_norm(h, w("group_norm2/gamma"), w("group_norm2/beta"), eps=AE_GN_EPS)),
             w("conv2/kernel"), w("conv2/bias"))
  if x.shape[-1] != h.shape[-1]:
    x = dense(x, w("shortcut/kernel"), w("shortcut/bias"))
  return h + x


def ae_attention_block(x, w):
  """autoencoder.py:74-97: GN -> q,k,v Dense (with bias) ->
  softmax(q.k^T * C**-0.5) over all H*W keys -> .v -> Dense -> + x."""
  b, hh, ww, c = x.shape
  h = group_norm(x, w("group_norm/gamma"), w("group_norm/beta"), eps=AE_GN_EPS)
  q = dense(h, w("dense_query/kernel"), w("dense_query/bias")).reshape(b, hh * ww, c)
  k = dense(h, w("dense_key/kernel"), w("dense_key/bias")).reshape(b, hh * ww, c)
  v = dense(h, w("dense_value/kernel"), w("dense_value/bias")).reshape(b, hh * ww, c)
  a = torch.softmax(torch.einsum("bqc,bkc->bqk", q, k) * (c ** -0.5), dim=-1)
  o = torch.einsum("bqk,bkc->bqc", a, v).reshape(b, hh, ww, c)
  return dense(o, w("dense_output/kernel"), w("dense_output/bias")) + x


def vq_nearest(latents, codebook):
  """quantize.py:57-78: d = |z|^2 + |e|^2 - 2 z.e^T, argmin, gather.  Returns
  (quantized, indices).  The straight-through form (:87) equals `quantized`
  numerically in the forward pass up to one rounding: z + (q - z)."""
  shp = latents.shape
  z = latents.reshape(-1, shp[-1])
  d = (z ** 2).sum(dim=1, keepdim=True) + (codebook ** 2).sum(dim=1) - 2 * (z @ codebook.t())
  idx = torch.argmin(d, dim=1)
  q = codebook[idx].reshape(shp)
  return latents + (q - latents), idx


def decoder_forward(latents, weights, dtype=torch.float32, attention_resolutions=(),
                    force_quantize=False):
  """AutoencoderKL.decode (autoencoder.py:361-364) / AutoencoderVQ.decode
  (:430-436, using element 0 of the quantizer's tuple -- the reference binds the
  whole tuple, a bug not reproduced) -> Decoder.call (:291-298).
  latents [B,h,w,4] (already divided by scale_factor) -> [B,8h,8w,3]."""
  w = W(weights, dtype)
  x = _t(latents, dtype)
  if force_quantize:
    x, _ = vq_nearest(x, w("quantize/kernel"))
  x = dense(x, w("post_quant_conv/kernel"), w("post_quant_conv/bias"))
  d = w.sub("decoder")
  h = conv2d(x, d("conv_in/kernel"), d("conv_in/bias"))
  h = ae_residual_block(h, d.sub("middle/residual1"))
  h = ae_attention_block(h, d.sub("middle/attention"))     # always (:193)
  h = ae_residual_block(h, d.sub("middle/residual2"))
  i = 0
  while d.has_prefix(f"up/{i}/"):
    u = d.sub(f"up/{i}")
    if u.has("conv/kernel"):                                # Upsample :149-156
      h = conv2d(upsample_nearest2x(h), u("conv/kernel"), u("conv/bias"))
    else:                                                   # UpBlock :173-178
      h = ae_residual_block(h, u.sub("residual"))
      if h.shape[1] in tuple(attention_resolutions):
        h = ae_attention_block(h, u.sub("attention"))
    i += 1
  h = silu(group_norm(h, d("group_norm/gamma"), d("group_norm/beta"), eps=AE_GN_EPS))
  return conv2d(h, d("conv_out/kernel"), d("conv_out/bias"))


def encoder_forward(images, weights, dtype=torch.float32, attention_resolutions=()):
  """Encoder.call (autoencoder.py:240-249) followed by quant_conv (:356 KL, :413 VQ).
  DownBlock = ResidualBlock (+ AttentionBlock when the run-time height is in
  `attention_resolutions`, :117); Downsample pads [[0,1],[0,1]] then 3x3 stride-2 VALID
  (:133-136).  images [B,H,W,3] -> [B,H/f,W/f,zc] (KL: zc = 2*latent moments)."""
  w = W(weights, dtype)
  e = w.sub("encoder")
  h = conv2d(_t(images, dtype), e("conv_in/kernel"), e("conv_in/bias"))
  i = 0
  while e.has_prefix(f"down/{i}/"):
    u = e.sub(f"down/{i}")
    if u.has("conv/kernel"):
      h = conv2d(h, u("conv/kernel"), u("conv/bias"), stride=2, pad=((0, 1), (0, 1)))
    else:
      h = ae_residual_block(h, u.sub("residual"))
      if h.shape[1] in tuple(attention_resolutions):
        h = ae_attention_block(h, u.sub("attention"))
    i += 1
  h = ae_residual_block(h, e.sub("middle/residual1"))
  h = ae_attention_block(h, e.sub("middle/attention"))
  h = ae_residual_block(h, e.sub("middle/residual2"))
  h = conv2d(silu(group_norm(h, e("group_norm/gamma"), e("group_norm/beta"), eps=AE_GN_EPS)),
             e("conv_out/kernel"), e("conv_out/bias"))
  return dense(h, w("quant_conv/kernel"), w("quant_conv/bias"))


def diagonal_gaussian(moments, noise=None):
  """distribution.py:6-25,50-51: mean, logvar = split(moments); the stored logvar is clipped
  to [-30, 20] but std = exp(0.5 * logvar) uses the UNCLIPPED value (:16-18, reproduced).
  Returns (mean, clipped logvar, sample) with sample = mean + std * noise (mode if None)."""
  mean, logvar = torch.chunk(moments, 2, dim=-1)
  std = torch.exp(0.5 * logvar)
  sample = mean if noise is None else mean + std * _t(noise, moments.dtype)
  return mean, torch.clamp(logvar, -30.0, 20.0), sample


def vq_encode(images, weights, dtype=torch.float32, attention_resolutions=(), beta=0.25):
  """AutoencoderVQ.encode (autoencoder.py:411-419): encoder + quant_conv, then the
  quantizer's (quantized, codebook_loss, indices) (quantize.py:57-90)."""
  z = encoder_forward(images, weights, dtype, attention_resolutions)
  q, idx = vq_nearest(z, W(weights, dtype)("quantize/kernel"))
  qq = W(weights, dtype)("quantize/kernel")[idx].reshape(z.shape)
  loss = ((qq - z) ** 2).mean() + beta * ((qq - z) ** 2).mean()
  return z, q, loss, idx


# ----------------------------------------------------------------------------
# model_runners.py : schedule, DDIM step, loop
# ----------------------------------------------------------------------------

def tf_linspace_f32(start, stop, num):
  """tf.linspace on float32 [TF-mem: math_ops.linspace_nd]: endpoints exact,
  interior = start + delta*i with delta=(stop-start)/(num-1), all in float32."""
  start = np.float32(start)
  stop = np.float32(stop)
  delta = np.float32((stop - start) / np.float32(num - 1))
  inner = start + delta * np.arange(1, num - 1, dtype=np.float32)
  return np.concatenate([[start], inner.astype(np.float32), [stop]]).astype(np.float32)


def make_schedule(num_steps=1000, beta_start=1e-4, beta_end=2e-2, eta=0.,
                  num_ddim_steps=50):
  """model_runners.py:379-423.  betas = f64(linspace_f32(sqrt b0, sqrt b1, T)**2)
  (:379-382; the Python-float sqrt is float64, the tensor float32); cumprod in
  f64; ddim_steps = range(0,T,T//N) (+1 if N<T) (:406-409); a_prev =
  [abar[0]] + abar[steps[:-1]] (:412-415); sigma (:416-419); c1, c2 (:420-423)."""
  ls = tf_linspace_f32(beta_start ** 0.5, beta_end ** 0.5, num_steps)
  betas = (ls * ls).astype(np.float32).astype(np.float64)
  alphas_cumprod = np.cumprod(1.0 - betas)
  steps = np.arange(0, num_steps, num_steps // num_ddim_steps, dtype=np.int32)
  if num_ddim_steps < num_steps:
    steps = steps + 1
  ac = alphas_cumprod[steps]          # out-of-range index raises, as tf.gather does on CPU
  ac_prev = np.concatenate([[alphas_cumprod[0]], alphas_cumprod[steps[:-1]]])
  sigmas = eta * np.sqrt((1 - ac_prev) / (1 - ac) * (1 - ac / ac_prev))
  return dict(
      ddim_steps=steps,
      alphas_cumprod=alphas_cumprod,
      ddim_alphas_cumprod_prev=ac_prev,
      ddim_sigmas=sigmas,
      ddim_sqrt_recip_alphas_cumprod=np.sqrt(1. / alphas_cumprod)[steps],
      ddim_sqrt_recipm1_alphas_cumprod=np.sqrt(1. / alphas_cumprod - 1)[steps],
  )


def ddim_update(xt, eps_uncond, eps_cond, sched, index, guidance_scale, noise,
                dtype=torch.float32, clip_denoised=False):
  """model_runners.py:453-468 given the two U-Net halves.  `_extract` casts the
  f64 tables to f32 BEFORE gathering (:41-44), so the sqrt at :463-464 runs in
  f32; in the f64 oracle the same f32-rounded scalars are used, widened."""
  f = lambda name: torch.tensor(np.float32(sched[name][index]), dtype=torch.float32)
  c1 = f("ddim_sqrt_recip_alphas_cumprod").to(dtype)
  c2 = f("ddim_sqrt_recipm1_alphas_cumprod").to(dtype)
  a_prev32 = f("ddim_alphas_cumprod_prev")
  std32 = f("ddim_sigmas")
  if dtype == torch.float32:
    sa = torch.sqrt(a_prev32)
    sb = torch.sqrt(1 - a_prev32 - std32 ** 2)
  else:
    sa = torch.sqrt(a_prev32.to(dtype))
    sb = torch.sqrt(1 - a_prev32.to(dtype) - std32.to(dtype) ** 2)
  eps = eps_uncond + guidance_scale * (eps_cond - eps_uncond)
  pred_x0 = c1 * xt - c2 * eps
  if clip_denoised:
    pred_x0 = pred_x0.clamp(-1, 1)
  mean = sa * pred_x0 + sb * eps
  return mean + _t(noise, dtype) * std32.to(dtype), pred_x0


def ddim_sample(xt, cond, index, sched, unet_weights, guidance_scale=1.,
                noise=None, dtype=torch.float32, clip_denoised=True, num_heads=8):
  """model_runners.py:438-472: t = fill([2B], steps[index]); ONE U-Net call on
  concat([xt, xt]); split; CFG; DDIM update."""
  xt = _t(xt, dtype)
  b = xt.shape[0]
  t = np.full([2 * b], sched["ddim_steps"][index], dtype=np.int32)
  eps_all = unet_forward(torch.cat([xt, xt], dim=0), t, cond, unet_weights, dtype, num_heads)
  if noise is None:
    noise = torch.zeros_like(xt)
  out, pred_x0 = ddim_update(xt, eps_all[:b], eps_all[b:], sched, index, guidance_scale,
                             noise, dtype, clip_denoised)
  return out, pred_x0, eps_all


def ddim_p_sample_loop(token_ids, x_T, weights, ldm, guidance_scale=5.,
                       noises=None, dtype=torch.float32, autoencoder_type="kl",
                       ae_attention_resolutions=(), record=None, num_heads=8):
  """model_runners.py:474-509.  `weights` = dict(unet=..., autoencoder=...,
  cond_stage_model=...); `ldm` = the YAML `ldm` section.  x_T and the per-step
  noises are explicit inputs (the reference draws unseeded tf.random.normal,
  :466,:478).  Returns images f[B,8h,8w,3]."""
  sched = make_schedule(ldm.get("num_steps", 1000), ldm.get("beta_start", 1e-4),
                        ldm.get("beta_end", 2e-2), ldm.get("eta", 0.),
                        ldm.get("num_ddim_steps", 50))
  context = text_encoder(token_ids, weights["cond_stage_model"], dtype)
  xt = _t(x_T, dtype)
  n = len(sched["ddim_steps"])
  for index in range(n - 1, -1, -1):
    noise = None if noises is None else noises[index]
    xt, _, _ = ddim_sample(xt, context, index, sched, weights["unet"], guidance_scale,
                           noise, dtype, clip_denoised=False, num_heads=num_heads)
    if record is not None:
      record.append(xt.clone())
  latents = xt / ldm.get("scale_factor", 0.18215)          # :426 division
  return decoder_forward(latents, weights["autoencoder"], dtype,
                         attention_resolutions=ae_attention_resolutions,
                         force_quantize=(autoencoder_type == "vq"))


def ddim_p_sample_loop_progressive(token_ids, x_T, weights, ldm, guidance_scale=5., record_freq=5,
                                   noises=None, dtype=torch.float32, num_heads=8):
  """Intended semantics of model_runners.py:511-575 (its two bugs -- the call to a
  non-existent `ddim_p_sample`, :535, and the 3-vs-2 value unpacking in the CLI -- not
  reproduced): slot r keeps sample / pred_x0 of the last step with index // record_freq == r
  (insert_mask, :545-553); all three results are decoded (:565-570)."""
  sched = make_schedule(ldm.get("num_steps", 1000), ldm.get("beta_start", 1e-4),
                        ldm.get("beta_end", 2e-2), ldm.get("eta", 0.), ldm.get("num_ddim_steps", 50))
  context = text_encoder(token_ids, weights["cond_stage_model"], dtype)
  xt = _t(x_T, dtype)
  n = len(sched["ddim_steps"])
  num_records = n // record_freq
  sp = torch.zeros((xt.shape[0], num_records) + tuple(xt.shape[1:]), dtype=dtype)
  xp = torch.zeros_like(sp)
  for index in range(n - 1, -1, -1):
    noise = None if noises is None else noises[index]
    xt, pred_x0, _ = ddim_sample(xt, context, index, sched, weights["unet"], guidance_scale, noise,
                                 dtype, clip_denoised=False, num_heads=num_heads)
    r = index // record_freq
    if r < num_records:
      sp[:, r] = xt
      xp[:, r] = pred_x0
  sf = ldm.get("scale_factor", 0.18215)
  dec = lambda z: decoder_forward(z / sf, weights["autoencoder"], dtype)
  b = xt.shape[0]
  images = dec(xt)
  spd = dec(sp.reshape((b * num_records,) + tuple(xt.shape[1:])))
  xpd = dec(xp.reshape((b * num_records,) + tuple(xt.shape[1:])))
  return images, spd.reshape((b, num_records) + tuple(spd.shape[1:])), xpd.reshape((b, num_records) + tuple(xpd.shape[1:]))


def tensor_to_image(images):
  """run_ldm_sampler.py:18-25: per-image (x-min)/(max-min), *255, astype(uint8)
  (truncation)."""
  a = np.array(images, dtype=np.float32, copy=True)
  for i in range(a.shape[0]):
    a[i] = (a[i] - a[i].min()) / (a[i].max() - a[i].min())
  a *= 255
  return a.astype("uint8")
