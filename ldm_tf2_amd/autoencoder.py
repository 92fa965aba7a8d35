"""First-stage autoencoders on the HIP path -- host side.

Mirrors `AutoencoderKL.decode` (autoencoder.py:361-364) and `AutoencoderVQ.decode`
(:430-436) with the reference constructors' kwargs (= YAML `autoencoder_kl` /
`autoencoder_vq`).  `decode(latents f32 [B,h,w,4]) -> f32 [B,8h,8w,3]`, NHWC.
`encode` (:353-359 KL, :411-419 VQ) is built when the model is
constructed with `with_encoder=True` (or handed encoder weights): images f32 [B,H,W,3] ->
DiagonalGaussian (KL) / (latents, codebook_loss, indices) (VQ).

VQ: the reference binds the quantizer's 3-tuple to `latents` (:432), which cannot
run; the intended element 0 (the quantised latents) is used.
"""
from __future__ import annotations

import torch

from . import layout as L
from . import ops
from .weights import decoder_manifest, encoder_manifest, init_weights

GROUP_NORM_EPS = 1e-6   # autoencoder.py:11


def _chunk_rows(batch, sample_elems, dtype, max_chunk):
  """Images per decoder / encoder pass: below ldm_gemm's 2 GiB operand limit and `max_chunk`."""
  esize = 2 if dtype == torch.bfloat16 else 4
  limit = max(1, ((1 << 31) - 4096) // (sample_elems * esize))
  return max(1, min(batch, limit, max_chunk))


class _Res:
  def __init__(self, w, p, dtype, dev):
    g = lambda n: w[p + "/" + n]
    self.cin, self.cout = g("conv1/kernel").shape[2], g("conv1/kernel").shape[3]
    self.gn1 = (L.vec(g("group_norm1/gamma"), dev), L.vec(g("group_norm1/beta"), dev))
    self.conv1 = (L.conv_kernel(g("conv1/kernel"), dtype, dev), L.vec(g("conv1/bias"), dev))
    self.gn2 = (L.vec(g("group_norm2/gamma"), dev), L.vec(g("group_norm2/beta"), dev))
    self.conv2 = (L.conv_kernel(g("conv2/kernel"), dtype, dev), L.vec(g("conv2/bias"), dev))
    self.shortcut = None
    if (p + "/shortcut/kernel") in w:
      self.shortcut = (L.dense_kernel(g("shortcut/kernel"), dtype, dev), L.vec(g("shortcut/bias"), dev))


class _Attn:
  def __init__(self, w, p, dtype, dev):
    g = lambda n: w[p + "/" + n]
    self.gn = (L.vec(g("group_norm/gamma"), dev), L.vec(g("group_norm/beta"), dev))
    d = lambda n: (L.dense_kernel(g(n + "/kernel"), dtype, dev), L.vec(g(n + "/bias"), dev))
    self.q, self.k, self.v, self.o = d("dense_query"), d("dense_key"), d("dense_value"), d("dense_output")
    self.c = g("dense_query/kernel").shape[0]


class _Blocks:
  """ResidualBlock / AttentionBlock launches shared by the decoder and the encoder."""

  def _res(self, r, x, out):
    """autoencoder.py:42-58 with time=None."""
    B_, dt = self.buf, self.dtype
    B, h, w, _ = x.shape
    t0 = B_.get("gn", (B, h, w, r.cin), dt)
    ops.groupnorm(x, r.gn1[0], r.gn1[1], t0, GROUP_NORM_EPS, silu=True, partial=self._gnp)
    h1 = B_.get("h1", (B, h, w, r.cout), dt)
    ops.conv3x3(t0, r.conv1[0], h1, bias=r.conv1[1])
    t1 = B_.get("gn", (B, h, w, r.cout), dt)
    ops.groupnorm(h1, r.gn2[0], r.gn2[1], t1, GROUP_NORM_EPS, silu=True, partial=self._gnp)
    res = x
    if r.shortcut is not None:
      res = B_.get("sc", (B, h, w, r.cout), dt)
      ops.linear(x, r.shortcut[0], res, bias=r.shortcut[1])
    ops.conv3x3(t1, r.conv2[0], out, bias=r.conv2[1], residual=res)
    return out

  # queries per pass of the single-head attention: logits are materialised for one block of queries
  # at a time, [B, ATTN_QBLOCK, T] f32 (33 MB at B=4, T=4096) instead of [B, T, T] (268 MB)
  ATTN_QBLOCK = 512

  def _attn(self, a, x, out):
    """autoencoder.py:74-97: single head over all H*W positions, head dim = C.  C = 512 (every real
    configuration) runs the fused wide-head attention kernel (attention.hip: attn_wide_kernel).
    Other widths (the tiny test configurations) fall back to q.k^T (batched MFMA GEMM, f32 out) ->
    row softmax with the C**-0.5 scale -> P.V (batched GEMM against V^T), ATTN_QBLOCK queries at
    a time: rows of a softmax are independent, so blocking the queries changes nothing but the
    size of the logits scratch."""
    B_, dt = self.buf, self.dtype
    B, h, w, c = x.shape
    T = h * w
    t0 = B_.get("gn", (B, h, w, c), dt)
    ops.groupnorm(x, a.gn[0], a.gn[1], t0, GROUP_NORM_EPS, silu=False, partial=self._gnp)
    q = B_.get("at_q", (B, T, c), dt)
    k = B_.get("at_k", (B, T, c), dt)
    vt = B_.get("at_vt", (B, c, T), dt)
    ops.linear(t0, a.q[0], q, bias=a.q[1])
    ops.linear(t0, a.k[0], k, bias=a.k[1])
    ops.bmm_nt(t0.reshape(B, T, c), a.v[0], vt, bias=a.v[1], transposed_out=True)
    o = B_.get("at_o", (B, T, c), dt)
    if c == 512:
      # the real configurations (KL-f8 / VQ-f8 mid block and VQ level-3 blocks: C = 512): fused
      # flash-style kernel with the head dim split over the four waves of a workgroup -- logits
      # never leave the chip
      ops.attention(q, k, vt, o, 1, 512, c ** -0.5)
      ops.linear(o, a.o[0], out, bias=a.o[1], residual=x)
      return out
    qb = max(d for d in range(1, min(T, self.ATTN_QBLOCK) + 1) if T % d == 0)   # whole blocks only
    logits = B_.get("at_logits", (B, qb, T), torch.float32)
    p = B_.get("at_p", (B, qb, T), dt)
    for q0 in range(0, T, qb):
      q1 = min(T, q0 + qb)
      n = q1 - q0
      ops.bmm_nt(q[:, q0:q1], k, logits[:, :n])
      ops.softmax_rows(logits[:, :n], p[:, :n], scale=c ** -0.5)
      ops.bmm_nt(p[:, :n], vt, o[:, q0:q1])
    ops.linear(o, a.o[0], out, bias=a.o[1], residual=x)
    return out

  def _dst(self, cur, shape):
    """Ping-pong activation buffer of `shape` that is not `cur`."""
    t = self.buf.get("act_a", shape, self.dtype)
    if cur is not None and t.data_ptr() == cur.data_ptr():
      t = self.buf.get("act_b", shape, self.dtype)
    return t


class _Decoder(_Blocks):
  """Decoder (autoencoder.py:252-298) + post_quant_conv (+ VQ codebook)."""

  def __init__(self, weights, dtype, device, attention_resolutions):
    w, dev = weights, device
    self.dtype, self.device = dtype, dev
    self.attention_resolutions = tuple(attention_resolutions)
    self.codebook = L.vec(w["quantize/kernel"], dev) if "quantize/kernel" in w else None
    self.post_quant = (L.vec(w["post_quant_conv/kernel"], dev), L.vec(w["post_quant_conv/bias"], dev))
    self.conv_in = (L.vec(w["decoder/conv_in/kernel"], dev), L.vec(w["decoder/conv_in/bias"], dev))
    self.mid = (_Res(w, "decoder/middle/residual1", dtype, dev),
                _Attn(w, "decoder/middle/attention", dtype, dev),
                _Res(w, "decoder/middle/residual2", dtype, dev))
    self.up = []
    i = 0
    while any(k.startswith(f"decoder/up/{i}/") for k in w):
      p = f"decoder/up/{i}"
      if (p + "/conv/kernel") in w:
        self.up.append(("up", L.conv_kernel(w[p + "/conv/kernel"], dtype, dev), L.vec(w[p + "/conv/bias"], dev)))
      else:
        a = _Attn(w, p + "/attention", dtype, dev) if (p + "/attention/group_norm/gamma") in w else None
        self.up.append(("res", _Res(w, p + "/residual", dtype, dev), a))
      i += 1
    self.gn_out = (L.vec(w["decoder/group_norm/gamma"], dev), L.vec(w["decoder/group_norm/beta"], dev))
    self.conv_out = (L.vec(w["decoder/conv_out/kernel"], dev), L.vec(w["decoder/conv_out/bias"], dev))
    self.buf = L.Buffers(dev)
    self._ws = ops.new_workspace(dev)
    self.max_chunk = 16

  def decode(self, latents, scale_factor=1.0, force_quantize=False):
    """latents f32 [B,h,w,C]; computes Decoder(post_quant(quantize?(latents / scale_factor))).
    Large batches run in chunks (samples are independent): no kernel operand may reach 2 GiB
    (ldm_gemm addresses operands with 32-bit buffer offsets) and the scratch stays bounded
    (`max_chunk` images at a time); the last chunk is the window ending at B, so every chunk
    has the same shape and reuses the same scratch buffers."""
    assert latents.dtype == torch.float32 and latents.is_contiguous()
    B, h, w, c = latents.shape
    f = 2 ** sum(1 for blk in self.up if blk[0] == "up")
    cout = self.conv_out[0].shape[-1]
    out = torch.empty(B, f * h, f * w, cout, dtype=torch.float32, device=self.device)
    n = _chunk_rows(B, self._max_sample_elems(h, w), self.dtype, self.max_chunk)
    with ops.workspace_scope(self._ws):
      for i in range(0, B, n):
        i0 = min(i, B - n)
        self._decode_chunk(latents[i0:i0 + n], scale_factor, force_quantize, out[i0:i0 + n])
    return out

  def _max_sample_elems(self, h, w):
    """Largest activation of one sample, in elements (conv inputs / outputs and their GroupNorm
    copies): channels x pixels at every level of the up path."""
    ch = self.mid[0].cin
    best = h * w * ch
    for blk in self.up:
      if blk[0] == "up":
        h, w = 2 * h, 2 * w
        ch = blk[1].shape[0]
      else:
        ch = max(blk[1].cin, blk[1].cout)
      best = max(best, h * w * ch)
    return best

  def _decode_chunk(self, latents, scale_factor, force_quantize, out):
    B_, dt = self.buf, self.dtype
    B, h, w, c = latents.shape
    self._gnp = B_.get("gn_partial", (B * 128 * 32 * 2,), torch.float32)
    z = latents
    sf = scale_factor
    if force_quantize:
      if self.codebook is None:
        raise ValueError("force_quantize needs a VQ codebook")
      zs = B_.get("zs", (B, h, w, c), torch.float32)
      # quantise latents/scale_factor: scale with an identity post_quant, then look up
      eye = B_.get("eye", (c, c), torch.float32, zero=True)
      if not getattr(self, "_eye_ok", False):
        eye.copy_(torch.eye(c))
        self._eye_ok = True
      ops.post_quant(z, sf, eye, None, zs)
      zq = B_.get("zq", (B, h, w, c), torch.float32)
      ops.vq_nearest(zs, self.codebook, zq)
      z, sf = zq, 1.0
    x0 = B_.get("pq", (B, h, w, c), torch.float32)
    ops.post_quant(z, sf, self.post_quant[0], self.post_quant[1], x0)
    ch = self.mid[0].cin
    cur = self._dst(None, (B, h, w, ch))
    ops.conv3x3_small(x0, self.conv_in[0], self.conv_in[1], cur)
    cur = self._res(self.mid[0], cur, self._dst(cur, (B, h, w, ch)))
    cur = self._attn(self.mid[1], cur, self._dst(cur, (B, h, w, ch)))   # always (autoencoder.py:193)
    cur = self._res(self.mid[2], cur, self._dst(cur, (B, h, w, ch)))
    for blk in self.up:
      hh, ww = cur.shape[1], cur.shape[2]
      if blk[0] == "up":
        dst = self._dst(cur, (B, 2 * hh, 2 * ww, blk[1].shape[0]))
        cur = ops.conv3x3(cur, blk[1], dst, bias=blk[2], upsample=True)   # autoencoder.py:152-155
      else:
        _, r, a = blk
        cur = self._res(r, cur, self._dst(cur, (B, hh, ww, r.cout)))
        if a is not None and hh in self.attention_resolutions:            # autoencoder.py:176
          cur = self._attn(a, cur, self._dst(cur, (B, hh, ww, r.cout)))
    t0 = B_.get("gn", tuple(cur.shape), dt)
    ops.groupnorm(cur, self.gn_out[0], self.gn_out[1], t0, GROUP_NORM_EPS, silu=True, partial=self._gnp)
    ops.conv3x3_small(t0, self.conv_out[0], self.conv_out[1], out)
    return out


class _Encoder(_Blocks):
  """Encoder (autoencoder.py:198-249) + quant_conv."""

  def __init__(self, weights, dtype, device, attention_resolutions):
    w, dev = weights, device
    self.dtype, self.device = dtype, dev
    self.attention_resolutions = tuple(attention_resolutions)
    self.conv_in = (L.vec(w["encoder/conv_in/kernel"], dev), L.vec(w["encoder/conv_in/bias"], dev))
    self.down = []
    i = 0
    while any(k.startswith(f"encoder/down/{i}/") for k in w):
      p = f"encoder/down/{i}"
      if (p + "/conv/kernel") in w:
        self.down.append(("down", L.conv_kernel(w[p + "/conv/kernel"], dtype, dev), L.vec(w[p + "/conv/bias"], dev)))
      else:
        a = _Attn(w, p + "/attention", dtype, dev) if (p + "/attention/group_norm/gamma") in w else None
        self.down.append(("res", _Res(w, p + "/residual", dtype, dev), a))
      i += 1
    self.mid = (_Res(w, "encoder/middle/residual1", dtype, dev),
                _Attn(w, "encoder/middle/attention", dtype, dev),
                _Res(w, "encoder/middle/residual2", dtype, dev))
    self.gn_out = (L.vec(w["encoder/group_norm/gamma"], dev), L.vec(w["encoder/group_norm/beta"], dev))
    self.conv_out = (L.conv_kernel(w["encoder/conv_out/kernel"], dtype, dev), L.vec(w["encoder/conv_out/bias"], dev))
    self.quant = (L.vec(w["quant_conv/kernel"], dev), L.vec(w["quant_conv/bias"], dev))
    self.zc = w["quant_conv/kernel"].shape[0]
    self.buf = L.Buffers(dev)
    self._ws = ops.new_workspace(dev)
    self.max_chunk = 16

  def encode(self, images):
    """images f32 [B,H,W,3] -> quant_conv(Encoder(images)) f32 [B,H/f,W/f,zc]; chunked over the
    batch like `_Decoder.decode`."""
    assert images.dtype == torch.float32 and images.is_contiguous()
    B, H, W, _ = images.shape
    f = 2 ** sum(1 for blk in self.down if blk[0] == "down")
    out = torch.empty(B, H // f, W // f, self.zc, dtype=torch.float32, device=self.device)
    hh, ww, per = H, W, H * W * self.conv_in[0].shape[-1]
    for blk in self.down:
      if blk[0] == "down":
        hh, ww = hh // 2, ww // 2
      else:
        per = max(per, hh * ww * max(blk[1].cin, blk[1].cout))
    n = _chunk_rows(B, per, self.dtype, self.max_chunk)
    with ops.workspace_scope(self._ws):
      for i in range(0, B, n):
        i0 = min(i, B - n)
        self._encode_chunk(images[i0:i0 + n], out[i0:i0 + n])
    return out

  def _encode_chunk(self, images, out):
    B_, dt = self.buf, self.dtype
    B, H, W, _ = images.shape
    self._gnp = B_.get("gn_partial", (B * 128 * 32 * 2,), torch.float32)
    ch = self.conv_in[0].shape[-1]
    cur = self._dst(None, (B, H, W, ch))
    ops.conv3x3_small(images, self.conv_in[0], self.conv_in[1], cur)
    for blk in self.down:
      hh, ww = cur.shape[1], cur.shape[2]
      if blk[0] == "down":                                               # autoencoder.py:133-136
        dst = self._dst(cur, (B, hh // 2, ww // 2, blk[1].shape[0]))
        cur = ops.conv3x3(cur, blk[1], dst, bias=blk[2], stride=2, no_lead_pad=True)
      else:
        _, r, a = blk
        cur = self._res(r, cur, self._dst(cur, (B, hh, ww, r.cout)))
        if a is not None and hh in self.attention_resolutions:            # autoencoder.py:117
          cur = self._attn(a, cur, self._dst(cur, (B, hh, ww, r.cout)))
    shp = tuple(cur.shape)
    cur = self._res(self.mid[0], cur, self._dst(cur, shp))
    cur = self._attn(self.mid[1], cur, self._dst(cur, shp))
    cur = self._res(self.mid[2], cur, self._dst(cur, shp))
    t0 = B_.get("gn", shp, dt)
    ops.groupnorm(cur, self.gn_out[0], self.gn_out[1], t0, GROUP_NORM_EPS, silu=True, partial=self._gnp)
    h = B_.get("enc_h", (B, shp[1], shp[2], self.zc), torch.float32)
    ops.conv3x3(t0, self.conv_out[0], h, bias=self.conv_out[1])
    ops.post_quant(h, 1.0, self.quant[0], self.quant[1], out)
    return out


class DiagonalGaussian:
  """distribution.py:6-51 (the members the encode path uses).  `sample(noise=None, seed=0)`:
  the reference draws tf.random.normal; here the noise is an explicit input or comes from a
  seeded torch generator."""

  def __init__(self, moments):
    self._moments = moments
    c = moments.shape[-1] // 2
    self._mean = moments[..., :c]
    self._logvar = torch.clamp(moments[..., c:], -30.0, 20.0)      # :16 (std uses the unclipped value, :18)
    self._shape = tuple(moments.shape[:-1]) + (c,)

  def mode(self):
    out = torch.empty(self._shape, dtype=torch.float32, device=self._moments.device)
    return ops.gaussian_sample(self._moments, out)

  def sample(self, noise=None, seed=0):
    dev = self._moments.device
    if noise is None:
      g = torch.Generator(device="cpu").manual_seed(int(seed))
      noise = torch.randn(self._shape, generator=g, dtype=torch.float32)
    noise = torch.as_tensor(noise, dtype=torch.float32).to(dev).contiguous()
    out = torch.empty(self._shape, dtype=torch.float32, device=dev)
    return ops.gaussian_sample(self._moments, out, noise=noise)


class _AutoencoderBase:
  _is_vq = False

  def _build(self, man_kwargs, weights, dtype, device, init, seed, attention_resolutions,
             with_encoder=None, enc_kwargs=None):
    self.dtype, self.device = dtype, torch.device(device)
    self.manifest = decoder_manifest(**man_kwargs)
    if with_encoder is None:
      with_encoder = weights is not None and "encoder/conv_in/kernel" in weights
    if with_encoder:
      self.manifest.update(encoder_manifest(**enc_kwargs))
    if weights is None:
      weights = init_weights(self.manifest, seed=seed, mode=init, scope="autoencoder")
    missing = [k for k in self.manifest if k not in weights]
    if missing:
      raise KeyError(f"autoencoder weights missing {len(missing)} tensors, e.g. {missing[:3]}")
    self._decoder = _Decoder(weights, dtype, self.device, attention_resolutions)
    self._encoder = _Encoder(weights, dtype, self.device, attention_resolutions) if with_encoder else None

  def _encode(self, inputs):
    if self._encoder is None:
      raise RuntimeError("this autoencoder was built without its encoder: pass with_encoder=True "
                         "(or weights that contain encoder/*)")
    x = torch.as_tensor(inputs, dtype=torch.float32).to(self.device).contiguous()
    return self._encoder.encode(x)


class AutoencoderKL(_AutoencoderBase):
  """kwargs of autoencoder.py:302-311.  `attention_resolutions` is accepted and
  ignored exactly as the reference does (it passes `()` to its Decoder, :339)."""

  def __init__(self, latent_channels=4, channels=128, num_blocks=2, attention_resolutions=(),
               dropout_rate=0., multipliers=(1, 2, 4, 4), resample_with_conv=True, *,
               weights=None, dtype=torch.float32, device="cuda:0", init="keras", seed=2,
               with_encoder=None, image_size=256):
    if not resample_with_conv:
      raise NotImplementedError("resample_with_conv=False is not on the sampling path")
    self._latent_channels, self._channels, self._num_blocks = latent_channels, channels, num_blocks
    self._multipliers = tuple(multipliers)
    self._build(dict(latent_channels=latent_channels, channels=channels, num_blocks=num_blocks,
                     multipliers=self._multipliers, attention_resolutions=()),
                weights, dtype, device, init, seed, (), with_encoder,
                dict(latent_channels=latent_channels, channels=channels, num_blocks=num_blocks,
                     multipliers=self._multipliers, attention_resolutions=(), image_size=image_size,
                     double_z=True))

  def encode(self, inputs, training=False):
    """autoencoder.py:353-359: images [B,H,W,3] -> DiagonalGaussian posterior."""
    return DiagonalGaussian(self._encode(inputs))

  def decode(self, inputs, training=False, scale_factor=1.0):
    """autoencoder.py:361-364."""
    x = torch.as_tensor(inputs, dtype=torch.float32).to(self.device).contiguous()
    return self._decoder.decode(x, scale_factor=scale_factor)

  def call(self, inputs, sample_posterior=True, training=False, noise=None, seed=0):
    """autoencoder.py:344-351: (decode(posterior.sample() | posterior.mode()), posterior).  The
    sample's noise is an explicit input / seeded (the reference draws tf.random.normal)."""
    posterior = self.encode(inputs, training=training)
    latents = posterior.sample(noise=noise, seed=seed) if sample_posterior else posterior.mode()
    return self.decode(latents, training=training), posterior

  __call__ = call


class AutoencoderVQ(_AutoencoderBase):
  """kwargs of autoencoder.py:371-383.  `latent_size` (build-only) is the spatial
  size at which the decoder will run: it decides which UpBlocks own attention
  weights (autoencoder.py:176 tests the run-time size)."""
  _is_vq = True

  def __init__(self, latent_channels=4, channels=128, num_blocks=2, dropout_rate=0,
               multipliers=(1, 2, 2, 4), resample_with_conv=True, attention_resolutions=(32,),
               vocab_size=16384, beta=0.25, *, latent_size=32, weights=None,
               dtype=torch.float32, device="cuda:0", init="keras", seed=2, with_encoder=None,
               image_size=None):
    if not resample_with_conv:
      raise NotImplementedError("resample_with_conv=False is not on the sampling path")
    self._latent_channels, self._channels, self._num_blocks = latent_channels, channels, num_blocks
    self._multipliers, self._vocab_size, self._beta = tuple(multipliers), vocab_size, beta
    self._attention_resolutions = tuple(attention_resolutions)
    self._build(dict(latent_channels=latent_channels, channels=channels, num_blocks=num_blocks,
                     multipliers=self._multipliers, attention_resolutions=self._attention_resolutions,
                     latent_size=latent_size, vocab_size=vocab_size),
                weights, dtype, device, init, seed, self._attention_resolutions, with_encoder,
                dict(latent_channels=latent_channels, channels=channels, num_blocks=num_blocks,
                     multipliers=self._multipliers, attention_resolutions=self._attention_resolutions,
                     image_size=image_size or latent_size * 2 ** (len(self._multipliers) - 1),
                     double_z=False))

  def encode(self, inputs, only_encode=False, training=False):
    """autoencoder.py:411-419: latents, or (quantized latents, codebook_loss, indices)."""
    z = self._encode(inputs)
    if only_encode:
      return z
    q = torch.empty_like(z)
    idx = torch.empty(z.numel() // z.shape[-1], dtype=torch.int64, device=z.device)
    ops.vq_nearest(z, self._decoder.codebook, q, indices=idx)
    # quantize.py:80-85 (forward value; a training-only scalar, formed with torch on the device)
    e = self._decoder.codebook[idx].reshape(z.shape)
    loss = ((e - z) ** 2).mean() * (1.0 + self._beta)
    return q, loss, idx

  def decode(self, latents, force_quantize=False, training=False, scale_factor=1.0):
    """autoencoder.py:430-436."""
    x = torch.as_tensor(latents, dtype=torch.float32).to(self.device).contiguous()
    return self._decoder.decode(x, scale_factor=scale_factor, force_quantize=force_quantize)

  def call(self, inputs, return_indices=False, training=False):
    """autoencoder.py:438-444: (reconstruction, codebook loss[, code indices])."""
    quantized, loss, indices = self.encode(inputs, training=training)
    outputs = self.decode(quantized, training=training)
    return (outputs, loss, indices) if return_indices else (outputs, loss)

  __call__ = call
