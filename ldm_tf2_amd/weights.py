"""Weight manifest + seeded random initialiser for the three sampling-path models.

The manifest lists every variable of the reference's Keras models in the
reference's own layouts (conv kernels HWIO, dense kernels [in, out], attention
projections [D, H, S] / [H, S, D]) and in the reference's variable-creation
order, which is the order `convert_ckpt_pytorch_to_tf2.py:23-304` fills them in:

* U-Net            -> `unet.py:51-138` (order: convert_ckpt_pytorch_to_tf2.py:73-232)
* text transformer -> `transformer.py:218-272` (order: :23-70)
* KL / VQ decoder  -> `autoencoder.py:252-298,361-364,430-436` (order: :235-304)

Master weights are float32 numpy arrays in these layouts; the HIP host
(`ldm_tf2_amd.unet` etc.) re-lays them out for the device ("compiles" them) at
model-build time.  Nothing here touches the GPU.

Initialisation modes
  "keras"  : what the reference gets when `expect_partial()` silently restores
             nothing: glorot-uniform kernels, zero biases,
             gamma=1 / beta=0, Embedding U(-0.05, 0.05).   [TF-mem]
  "random" : as "keras" but biases, gammas and betas are random too, so parity
             tests exercise every affine/bias path.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

# ----------------------------------------------------------------------------
# manifests: OrderedDict name -> (shape, kind)
# kind in {"kernel", "bias", "gamma", "beta", "embedding", "codebook"}
# ----------------------------------------------------------------------------


def _resblock(m, p, cin, cout, temb_dim=None):
  m[p + "/group_norm_1/gamma"] = ((cin,), "gamma")
  m[p + "/group_norm_1/beta"] = ((cin,), "beta")
  m[p + "/conv2d_1/kernel"] = ((3, 3, cin, cout), "kernel")
  m[p + "/conv2d_1/bias"] = ((cout,), "bias")
  if temb_dim is not None:
    m[p + "/dense/kernel"] = ((temb_dim, cout), "kernel")
    m[p + "/dense/bias"] = ((cout,), "bias")
  m[p + "/group_norm_2/gamma"] = ((cout,), "gamma")
  m[p + "/group_norm_2/beta"] = ((cout,), "beta")
  m[p + "/conv2d_2/kernel"] = ((3, 3, cout, cout), "kernel")
  m[p + "/conv2d_2/bias"] = ((cout,), "bias")
  if cin != cout:
    m[p + "/shortcut/kernel"] = ((cin, cout), "kernel")
    m[p + "/shortcut/bias"] = ((cout,), "bias")


def _cross_attention(m, p, dq, dc, heads, sph):
  m[p + "/query/kernel"] = ((dq, heads, sph), "kernel")
  m[p + "/key/kernel"] = ((dc, heads, sph), "kernel")
  m[p + "/value/kernel"] = ((dc, heads, sph), "kernel")
  m[p + "/output/kernel"] = ((heads, sph, heads * sph), "kernel")
  m[p + "/output/bias"] = ((heads * sph,), "bias")


def _spatial_transformer(m, p, heads, sph, ctx_dim):
  c = heads * sph
  m[p + "/dense1/kernel"] = ((c, c), "kernel")
  m[p + "/dense1/bias"] = ((c,), "bias")
  _cross_attention(m, p + "/block/att_layer1", c, c, heads, sph)
  _cross_attention(m, p + "/block/att_layer2", c, ctx_dim, heads, sph)
  m[p + "/block/ffn/geglu/kernel"] = ((c, 8 * c), "kernel")
  m[p + "/block/ffn/geglu/bias"] = ((8 * c,), "bias")
  m[p + "/block/ffn/dense/kernel"] = ((4 * c, c), "kernel")
  m[p + "/block/ffn/dense/bias"] = ((c,), "bias")
  for i in (1, 2, 3):
    m[p + f"/block/layernorm{i}/gamma"] = ((c,), "gamma")
    m[p + f"/block/layernorm{i}/beta"] = ((c,), "beta")
  m[p + "/dense2/kernel"] = ((c, c), "kernel")
  m[p + "/dense2/bias"] = ((c,), "bias")
  m[p + "/groupnorm/gamma"] = ((c,), "gamma")
  m[p + "/groupnorm/beta"] = ((c,), "beta")


def unet_manifest(model_channels=320, out_channels=4, num_blocks=2,
                  channel_mult=(1, 2, 4, 4), num_heads=8, in_channels=4,
                  context_dim=1280, **_unused):
  """Variables of `UNet` (unet.py:51-116).  `size_per_head = 40*mult` in the
  reference (unet.py:82) == model_channels*mult/num_heads for its only valid
  configuration (320 channels, 8 heads); the second form is used so that small
  test configurations stay self-consistent (unet.py:360 reshapes heads*size
  back to channels)."""
  m = OrderedDict()
  mc = model_channels
  temb = 4 * mc
  m["conv_in/kernel"] = ((3, 3, in_channels, mc), "kernel")
  m["conv_in/bias"] = ((mc,), "bias")
  m["time_dense1/kernel"] = ((mc, temb), "kernel")
  m["time_dense1/bias"] = ((temb,), "bias")
  m["time_dense2/kernel"] = ((temb, temb), "kernel")
  m["time_dense2/bias"] = ((temb,), "bias")
  nlev = len(channel_mult)
  skips = [mc]
  ch = mc
  bi = 0
  for i, mult in enumerate(channel_mult):
    for _ in range(num_blocks):
      p = f"input_blocks/{bi}"
      _resblock(m, p + "/residual", ch, mc * mult, temb)
      ch = mc * mult
      if i < nlev - 1:
        _spatial_transformer(m, p + "/spatial_transformer", num_heads,
                             ch // num_heads, context_dim)
      skips.append(ch)
      bi += 1
    if i < nlev - 1:
      p = f"input_blocks/{bi}"
      m[p + "/downsample/conv/kernel"] = ((3, 3, ch, ch), "kernel")
      m[p + "/downsample/conv/bias"] = ((ch,), "bias")
      skips.append(ch)
      bi += 1
  _resblock(m, "middle_block/residual1", ch, ch, temb)
  _spatial_transformer(m, "middle_block/spatial_transformer", num_heads,
                       ch // num_heads, context_dim)
  _resblock(m, "middle_block/residual2", ch, ch, temb)
  bi = 0
  for i, mult in list(enumerate(channel_mult))[::-1]:
    for j in range(num_blocks + 1):
      p = f"output_blocks/{bi}"
      cin = ch + skips.pop()
      _resblock(m, p + "/residual", cin, mc * mult, temb)
      ch = mc * mult
      if i < nlev - 1:
        _spatial_transformer(m, p + "/spatial_transformer", num_heads,
                             ch // num_heads, context_dim)
      if i > 0 and j == num_blocks:
        m[p + "/upsample/conv/kernel"] = ((3, 3, ch, ch), "kernel")
        m[p + "/upsample/conv/bias"] = ((ch,), "bias")
      bi += 1
  m["groupnorm/gamma"] = ((ch,), "gamma")
  m["groupnorm/beta"] = ((ch,), "beta")
  m["conv_out/kernel"] = ((3, 3, ch, out_channels), "kernel")
  m["conv_out/bias"] = ((out_channels,), "bias")
  return m


def transformer_manifest(vocab_size=30522, encoder_stack_size=32,
                         hidden_size=1280, num_heads=8, size_per_head=64,
                         max_seq_len=77, filter_size=5120, **_unused):
  """Variables of `TransformerModel` that its `call` creates
  (transformer.py:218-272; `_logits_layer` is never built, :251)."""
  m = OrderedDict()
  d = hidden_size
  for i in range(encoder_stack_size):
    p = f"encoder/layers/{i}"
    m[p + "/mha/query/kernel"] = ((d, num_heads, size_per_head), "kernel")
    m[p + "/mha/key/kernel"] = ((d, num_heads, size_per_head), "kernel")
    m[p + "/mha/value/kernel"] = ((d, num_heads, size_per_head), "kernel")
    m[p + "/mha/output/kernel"] = ((num_heads, size_per_head, d), "kernel")
    m[p + "/mha/output/bias"] = ((d,), "bias")
    m[p + "/layernorm_mha/gamma"] = ((d,), "gamma")
    m[p + "/layernorm_mha/beta"] = ((d,), "beta")
    m[p + "/ffn/filter/kernel"] = ((d, filter_size), "kernel")
    m[p + "/ffn/filter/bias"] = ((filter_size,), "bias")
    m[p + "/ffn/output/kernel"] = ((filter_size, d), "kernel")
    m[p + "/ffn/output/bias"] = ((d,), "bias")
    m[p + "/layernorm_ffn/gamma"] = ((d,), "gamma")
    m[p + "/layernorm_ffn/beta"] = ((d,), "beta")
  m["encoder/layernorm/gamma"] = ((d,), "gamma")
  m["encoder/layernorm/beta"] = ((d,), "beta")
  m["embedding"] = ((vocab_size, d), "embedding")
  m["positional_embedding"] = ((max_seq_len, d), "embedding")
  return m


def _ae_resblock(m, p, cin, cout):
  # autoencoder.py:13-58 : no time embedding on the decode path (time=None)
  m[p + "/group_norm1/gamma"] = ((cin,), "gamma")
  m[p + "/group_norm1/beta"] = ((cin,), "beta")
  m[p + "/conv1/kernel"] = ((3, 3, cin, cout), "kernel")
  m[p + "/conv1/bias"] = ((cout,), "bias")
  m[p + "/group_norm2/gamma"] = ((cout,), "gamma")
  m[p + "/group_norm2/beta"] = ((cout,), "beta")
  m[p + "/conv2/kernel"] = ((3, 3, cout, cout), "kernel")
  m[p + "/conv2/bias"] = ((cout,), "bias")
  if cin != cout:
    m[p + "/shortcut/kernel"] = ((cin, cout), "kernel")
    m[p + "/shortcut/bias"] = ((cout,), "bias")


def _ae_attention(m, p, c):
  m[p + "/group_norm/gamma"] = ((c,), "gamma")
  m[p + "/group_norm/beta"] = ((c,), "beta")
  for n in ("dense_query", "dense_key", "dense_value", "dense_output"):
    m[p + f"/{n}/kernel"] = ((c, c), "kernel")
    m[p + f"/{n}/bias"] = ((c,), "bias")


def decoder_manifest(latent_channels=4, channels=128, num_blocks=2,
                     multipliers=(1, 2, 4, 4), attention_resolutions=(),
                     latent_size=32, out_channels=3, vocab_size=None,
                     **_unused):
  """Variables on the decode path of AutoencoderKL / AutoencoderVQ
  (autoencoder.py:252-298, :361-364, :430-436).  `attention_resolutions` only
  matters for VQ (KL hard-codes `()`, autoencoder.py:339); whether an UpBlock
  attention exists depends on the run-time spatial size (autoencoder.py:176),
  hence `latent_size`.  `vocab_size` adds the VQ codebook (quantize.py:34-39)."""
  m = OrderedDict()
  if vocab_size is not None:
    m["quantize/kernel"] = ((vocab_size, latent_channels), "codebook")
  m["post_quant_conv/kernel"] = ((latent_channels, latent_channels), "kernel")
  m["post_quant_conv/bias"] = ((latent_channels,), "bias")
  cl = [channels * mul for mul in multipliers]
  ch = cl[-1]
  m["decoder/conv_in/kernel"] = ((3, 3, latent_channels, ch), "kernel")
  m["decoder/conv_in/bias"] = ((ch,), "bias")
  _ae_resblock(m, "decoder/middle/residual1", ch, ch)
  _ae_attention(m, "decoder/middle/attention", ch)
  _ae_resblock(m, "decoder/middle/residual2", ch, ch)
  size = latent_size
  ui = 0
  for i in reversed(range(len(multipliers))):
    for _ in range(num_blocks + 1):
      p = f"decoder/up/{ui}"
      _ae_resblock(m, p + "/residual", ch, cl[i])
      ch = cl[i]
      if size in tuple(attention_resolutions):
        _ae_attention(m, p + "/attention", ch)
      ui += 1
    if i > 0:
      p = f"decoder/up/{ui}"
      m[p + "/conv/kernel"] = ((3, 3, ch, ch), "kernel")
      m[p + "/conv/bias"] = ((ch,), "bias")
      size *= 2
      ui += 1
  m["decoder/group_norm/gamma"] = ((ch,), "gamma")
  m["decoder/group_norm/beta"] = ((ch,), "beta")
  m["decoder/conv_out/kernel"] = ((3, 3, ch, out_channels), "kernel")
  m["decoder/conv_out/bias"] = ((out_channels,), "bias")
  return m


def encoder_manifest(latent_channels=4, channels=128, num_blocks=2,
                     multipliers=(1, 2, 4, 4), attention_resolutions=(),
                     image_size=256, in_channels=3, double_z=True, **_unused):
  """Variables on the encode path: Encoder
  (autoencoder.py:198-249) + quant_conv (:330 KL, :406 VQ).  KL: the encoder emits
  2*latent_channels moments (`double_z`, autoencoder.py:322) and never owns attention in
  its DownBlocks (:325); VQ: latent_channels and attention where the run-time size is in
  `attention_resolutions` (autoencoder.py:117), hence `image_size`."""
  m = OrderedDict()
  zc = latent_channels * (2 if double_z else 1)
  cl = [channels * mul for mul in multipliers]
  m["encoder/conv_in/kernel"] = ((3, 3, in_channels, channels), "kernel")
  m["encoder/conv_in/bias"] = ((channels,), "bias")
  ch, size, di = channels, image_size, 0
  for i in range(len(multipliers)):
    for _ in range(num_blocks):
      p = f"encoder/down/{di}"
      _ae_resblock(m, p + "/residual", ch, cl[i])
      ch = cl[i]
      if size in tuple(attention_resolutions):
        _ae_attention(m, p + "/attention", ch)
      di += 1
    if i < len(multipliers) - 1:
      p = f"encoder/down/{di}"
      m[p + "/conv/kernel"] = ((3, 3, ch, ch), "kernel")
      m[p + "/conv/bias"] = ((ch,), "bias")
      size //= 2
      di += 1
  _ae_resblock(m, "encoder/middle/residual1", ch, ch)
  _ae_attention(m, "encoder/middle/attention", ch)
  _ae_resblock(m, "encoder/middle/residual2", ch, ch)
  m["encoder/group_norm/gamma"] = ((ch,), "gamma")
  m["encoder/group_norm/beta"] = ((ch,), "beta")
  m["encoder/conv_out/kernel"] = ((3, 3, ch, zc), "kernel")
  m["encoder/conv_out/bias"] = ((zc,), "bias")
  m["quant_conv/kernel"] = ((zc, zc), "kernel")
  m["quant_conv/bias"] = ((zc,), "bias")
  return m


def count_params(manifest):
  return int(sum(int(np.prod(s)) for s, _ in manifest.values()))


# ----------------------------------------------------------------------------
# initialiser
# ----------------------------------------------------------------------------

def _fans(shape):
  """Keras `_compute_fans` [TF-mem]: rank-2 -> (in, out); rank>2 -> receptive
  field = prod(shape[:-2]), fan_in = shape[-2]*rf, fan_out = shape[-1]*rf."""
  if len(shape) == 1:
    return shape[0], shape[0]
  if len(shape) == 2:
    return shape[0], shape[1]
  rf = int(np.prod(shape[:-2]))
  return shape[-2] * rf, shape[-1] * rf


def init_weights(manifest, seed=2, mode="keras", scope=""):
  """Deterministic float32 weights for `manifest`.

  Every tensor draws from its own generator keyed by (seed, crc32(scope/name)),
  so a tensor's values do not depend on which other tensors exist."""
  assert mode in ("keras", "random")
  out = OrderedDict()
  for name, (shape, kind) in manifest.items():
    key = zlib.crc32((scope + "/" + name).encode())
    rng = np.random.default_rng([int(seed), key])
    if kind in ("kernel", "codebook"):
      fi, fo = _fans(shape)
      lim = np.float32(np.sqrt(6.0 / (fi + fo)))
      a = (rng.random(size=shape, dtype=np.float32) * np.float32(2.0) - np.float32(1.0)) * lim
    elif kind == "embedding":
      a = (rng.random(size=shape, dtype=np.float32) * np.float32(2.0) - np.float32(1.0)) * np.float32(0.05)
    elif kind == "bias":
      a = np.zeros(shape) if mode == "keras" else rng.uniform(-0.1, 0.1, size=shape)
    elif kind == "gamma":
      a = np.ones(shape) if mode == "keras" else rng.uniform(0.7, 1.3, size=shape)
    elif kind == "beta":
      a = np.zeros(shape) if mode == "keras" else rng.uniform(-0.2, 0.2, size=shape)
    else:
      raise ValueError(kind)
    out[name] = np.ascontiguousarray(a, dtype=np.float32)
  return out


def sub(weights, prefix):
  """View of `weights` restricted to names under `prefix/`, prefix stripped."""
  p = prefix.rstrip("/") + "/"
  return {k[len(p):]: v for k, v in weights.items() if k.startswith(p)}
