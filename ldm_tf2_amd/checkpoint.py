"""Real-weight loader.

Maps a CompVis latent-diffusion `state_dict` (the PyTorch checkpoint the reference converts
with convert_ckpt_pytorch_to_tf2.py) straight into this build's weight manifests
(`weights.py`: the reference's Keras variable names and layouts), applying the layout
changes that converter specifies:
  conv OIHW -> HWIO `.transpose(2,3,1,0)` (:78), Linear `.T` (:80), 1x1 conv `.squeeze().T`
  (:104), attention `.T.reshape(D,8,S)` (:111) / `.T.reshape(8,S,D)` (:114).
No checkpoint is reachable from the build or GPU boxes (no network), so this module is
exercised on synthetic state dicts of the right names/shapes (tests/test_host_cpu.py); the
name tables below are the converter's, expressed as rules instead of ordered lists.

    sd = torch.load("model.ckpt")["state_dict"]
    w = from_compvis_state_dict({k: v.numpy() for k, v in sd.items()})
    unet = UNet(**cfg["unet"], weights=w["unet"]); ...
"""
from __future__ import annotations

import numpy as np

from . import weights as Wt

_T = "model.diffusion_model."
_C = "cond_stage_model.transformer."
_D = "first_stage_model.decoder."


def _conv(a):        # OIHW -> HWIO
  return np.ascontiguousarray(np.asarray(a).transpose(2, 3, 1, 0))


def _lin(a):         # [out, in] -> [in, out]
  return np.ascontiguousarray(np.asarray(a).T)


def _c1(a):          # 1x1 conv [O, I, 1, 1] -> [I, O]
  a = np.asarray(a)
  return np.ascontiguousarray(a.reshape(a.shape[0], a.shape[1]).T)


def _split(heads):   # to_q/k/v [H*S, D] -> [D, H, S]
  fn = lambda a: np.ascontiguousarray(np.asarray(a).T.reshape(np.asarray(a).shape[1], heads, -1))
  fn._kind = "split"
  return fn


def _merge(heads):   # to_out [D, H*S] -> [H, S, D]
  fn = lambda a: np.ascontiguousarray(np.asarray(a).T.reshape(heads, -1, np.asarray(a).shape[0]))
  fn._kind = "merge"
  return fn


_ID = lambda a: np.ascontiguousarray(np.asarray(a))


def _res_rules(ours, theirs, rules, temb=True, ae=False):
  n1, n2 = ("group_norm1", "group_norm2") if ae else ("group_norm_1", "group_norm_2")
  c1, c2 = ("conv1", "conv2") if ae else ("conv2d_1", "conv2d_2")
  t = (("norm1", "conv1", "norm2", "conv2", "nin_shortcut") if ae else
       ("in_layers.0", "in_layers.2", "out_layers.0", "out_layers.3", "skip_connection"))
  rules[f"{ours}/{n1}/gamma"] = (f"{theirs}.{t[0]}.weight", _ID)
  rules[f"{ours}/{n1}/beta"] = (f"{theirs}.{t[0]}.bias", _ID)
  rules[f"{ours}/{c1}/kernel"] = (f"{theirs}.{t[1]}.weight", _conv)
  rules[f"{ours}/{c1}/bias"] = (f"{theirs}.{t[1]}.bias", _ID)
  if temb:
    rules[f"{ours}/dense/kernel"] = (f"{theirs}.emb_layers.1.weight", _lin)
    rules[f"{ours}/dense/bias"] = (f"{theirs}.emb_layers.1.bias", _ID)
  rules[f"{ours}/{n2}/gamma"] = (f"{theirs}.{t[2]}.weight", _ID)
  rules[f"{ours}/{n2}/beta"] = (f"{theirs}.{t[2]}.bias", _ID)
  rules[f"{ours}/{c2}/kernel"] = (f"{theirs}.{t[3]}.weight", _conv)
  rules[f"{ours}/{c2}/bias"] = (f"{theirs}.{t[3]}.bias", _ID)
  rules[f"{ours}/shortcut/kernel"] = (f"{theirs}.{t[4]}.weight", _c1)
  rules[f"{ours}/shortcut/bias"] = (f"{theirs}.{t[4]}.bias", _ID)


def _st_rules(ours, theirs, rules, heads):
  tb = f"{theirs}.transformer_blocks.0"
  rules[f"{ours}/dense1/kernel"] = (f"{theirs}.proj_in.weight", _c1)
  rules[f"{ours}/dense1/bias"] = (f"{theirs}.proj_in.bias", _ID)
  for ours_a, theirs_a in (("att_layer1", "attn1"), ("att_layer2", "attn2")):
    rules[f"{ours}/block/{ours_a}/query/kernel"] = (f"{tb}.{theirs_a}.to_q.weight", _split(heads))
    rules[f"{ours}/block/{ours_a}/key/kernel"] = (f"{tb}.{theirs_a}.to_k.weight", _split(heads))
    rules[f"{ours}/block/{ours_a}/value/kernel"] = (f"{tb}.{theirs_a}.to_v.weight", _split(heads))
    rules[f"{ours}/block/{ours_a}/output/kernel"] = (f"{tb}.{theirs_a}.to_out.0.weight", _merge(heads))
    rules[f"{ours}/block/{ours_a}/output/bias"] = (f"{tb}.{theirs_a}.to_out.0.bias", _ID)
  rules[f"{ours}/block/ffn/geglu/kernel"] = (f"{tb}.ff.net.0.proj.weight", _lin)
  rules[f"{ours}/block/ffn/geglu/bias"] = (f"{tb}.ff.net.0.proj.bias", _ID)
  rules[f"{ours}/block/ffn/dense/kernel"] = (f"{tb}.ff.net.2.weight", _lin)
  rules[f"{ours}/block/ffn/dense/bias"] = (f"{tb}.ff.net.2.bias", _ID)
  for i in (1, 2, 3):
    rules[f"{ours}/block/layernorm{i}/gamma"] = (f"{tb}.norm{i}.weight", _ID)
    rules[f"{ours}/block/layernorm{i}/beta"] = (f"{tb}.norm{i}.bias", _ID)
  rules[f"{ours}/dense2/kernel"] = (f"{theirs}.proj_out.weight", _c1)
  rules[f"{ours}/dense2/bias"] = (f"{theirs}.proj_out.bias", _ID)
  rules[f"{ours}/groupnorm/gamma"] = (f"{theirs}.norm.weight", _ID)
  rules[f"{ours}/groupnorm/beta"] = (f"{theirs}.norm.bias", _ID)


def unet_rules(manifest, heads=8):
  """manifest name -> (CompVis key, transform).  Block numbering as in
  convert_ckpt_pytorch_to_tf2.py:73-232 (CompVis input_blocks.0 is conv_in, so our input
  block i is theirs i+1; an output block's upsample sits after its transformer if any)."""
  r = {"conv_in/kernel": (_T + "input_blocks.0.0.weight", _conv), "conv_in/bias": (_T + "input_blocks.0.0.bias", _ID),
       "time_dense1/kernel": (_T + "time_embed.0.weight", _lin), "time_dense1/bias": (_T + "time_embed.0.bias", _ID),
       "time_dense2/kernel": (_T + "time_embed.2.weight", _lin), "time_dense2/bias": (_T + "time_embed.2.bias", _ID),
       "groupnorm/gamma": (_T + "out.0.weight", _ID), "groupnorm/beta": (_T + "out.0.bias", _ID),
       "conv_out/kernel": (_T + "out.2.weight", _conv), "conv_out/bias": (_T + "out.2.bias", _ID)}
  blocks = sorted({int(k.split("/")[1]) for k in manifest if k.startswith("input_blocks/")})
  for i in blocks:
    ours, theirs = f"input_blocks/{i}", f"{_T}input_blocks.{i + 1}"
    r[f"{ours}/downsample/conv/kernel"] = (f"{theirs}.0.op.weight", _conv)
    r[f"{ours}/downsample/conv/bias"] = (f"{theirs}.0.op.bias", _ID)
    _res_rules(f"{ours}/residual", f"{theirs}.0", r)
    _st_rules(f"{ours}/spatial_transformer", f"{theirs}.1", r, heads)
  _res_rules("middle_block/residual1", _T + "middle_block.0", r)
  _st_rules("middle_block/spatial_transformer", _T + "middle_block.1", r, heads)
  _res_rules("middle_block/residual2", _T + "middle_block.2", r)
  blocks = sorted({int(k.split("/")[1]) for k in manifest if k.startswith("output_blocks/")})
  for i in blocks:
    ours, theirs = f"output_blocks/{i}", f"{_T}output_blocks.{i}"
    _res_rules(f"{ours}/residual", f"{theirs}.0", r)
    has_st = any(k.startswith(f"{ours}/spatial_transformer/") for k in manifest)
    _st_rules(f"{ours}/spatial_transformer", f"{theirs}.1", r, heads)
    up = 2 if has_st else 1
    r[f"{ours}/upsample/conv/kernel"] = (f"{theirs}.{up}.conv.weight", _conv)
    r[f"{ours}/upsample/conv/bias"] = (f"{theirs}.{up}.conv.bias", _ID)
  return {k: v for k, v in r.items() if k in manifest}


def transformer_rules(manifest, heads=8):
  """convert_ckpt_pytorch_to_tf2.py:23-70: attention layer i is attn_layers.layers.{2i},
  its feed-forward {2i+1}; index .0 of each is the pre-LayerNorm, .1 the module."""
  r = {"encoder/layernorm/gamma": (_C + "norm.weight", _ID), "encoder/layernorm/beta": (_C + "norm.bias", _ID),
       "embedding": (_C + "token_emb.weight", _ID), "positional_embedding": (_C + "pos_emb.emb.weight", _ID)}
  n = len({k.split("/")[2] for k in manifest if k.startswith("encoder/layers/")})
  for i in range(n):
    o, a, f = f"encoder/layers/{i}", f"{_C}attn_layers.layers.{2 * i}", f"{_C}attn_layers.layers.{2 * i + 1}"
    r[f"{o}/mha/query/kernel"] = (f"{a}.1.to_q.weight", _split(heads))
    r[f"{o}/mha/key/kernel"] = (f"{a}.1.to_k.weight", _split(heads))
    r[f"{o}/mha/value/kernel"] = (f"{a}.1.to_v.weight", _split(heads))
    r[f"{o}/mha/output/kernel"] = (f"{a}.1.to_out.weight", _merge(heads))
    r[f"{o}/mha/output/bias"] = (f"{a}.1.to_out.bias", _ID)
    r[f"{o}/layernorm_mha/gamma"] = (f"{a}.0.weight", _ID)
    r[f"{o}/layernorm_mha/beta"] = (f"{a}.0.bias", _ID)
    r[f"{o}/ffn/filter/kernel"] = (f"{f}.1.net.0.0.weight", _lin)
    r[f"{o}/ffn/filter/bias"] = (f"{f}.1.net.0.0.bias", _ID)
    r[f"{o}/ffn/output/kernel"] = (f"{f}.1.net.2.weight", _lin)
    r[f"{o}/ffn/output/bias"] = (f"{f}.1.net.2.bias", _ID)
    r[f"{o}/layernorm_ffn/gamma"] = (f"{f}.0.weight", _ID)
    r[f"{o}/layernorm_ffn/beta"] = (f"{f}.0.bias", _ID)
  return {k: v for k, v in r.items() if k in manifest}


def decoder_rules(manifest, num_blocks=2, num_levels=4):
  """convert_ckpt_pytorch_to_tf2.py:235-304 (+ :419-423 for post_quant_conv)."""
  r = {"post_quant_conv/kernel": ("first_stage_model.post_quant_conv.weight", _c1),
       "post_quant_conv/bias": ("first_stage_model.post_quant_conv.bias", _ID),
       "quantize/kernel": ("first_stage_model.quantize.embedding.weight", _ID),
       "decoder/conv_in/kernel": (_D + "conv_in.weight", _conv), "decoder/conv_in/bias": (_D + "conv_in.bias", _ID),
       "decoder/group_norm/gamma": (_D + "norm_out.weight", _ID), "decoder/group_norm/beta": (_D + "norm_out.bias", _ID),
       "decoder/conv_out/kernel": (_D + "conv_out.weight", _conv), "decoder/conv_out/bias": (_D + "conv_out.bias", _ID)}

  def attn(ours, theirs):
    r[f"{ours}/group_norm/gamma"] = (f"{theirs}.norm.weight", _ID)
    r[f"{ours}/group_norm/beta"] = (f"{theirs}.norm.bias", _ID)
    for o, t in (("dense_query", "q"), ("dense_key", "k"), ("dense_value", "v"), ("dense_output", "proj_out")):
      r[f"{ours}/{o}/kernel"] = (f"{theirs}.{t}.weight", _c1)
      r[f"{ours}/{o}/bias"] = (f"{theirs}.{t}.bias", _ID)

  _res_rules("decoder/middle/residual1", _D + "mid.block_1", r, temb=False, ae=True)
  attn("decoder/middle/attention", _D + "mid.attn_1")
  _res_rules("decoder/middle/residual2", _D + "mid.block_2", r, temb=False, ae=True)
  ui = 0
  for lvl in reversed(range(num_levels)):
    for j in range(num_blocks + 1):
      _res_rules(f"decoder/up/{ui}/residual", f"{_D}up.{lvl}.block.{j}", r, temb=False, ae=True)
      attn(f"decoder/up/{ui}/attention", f"{_D}up.{lvl}.attn.{j}")
      ui += 1
    if lvl > 0:
      r[f"decoder/up/{ui}/conv/kernel"] = (f"{_D}up.{lvl}.upsample.conv.weight", _conv)
      r[f"decoder/up/{ui}/conv/bias"] = (f"{_D}up.{lvl}.upsample.conv.bias", _ID)
      ui += 1
  return {k: v for k, v in r.items() if k in manifest}


def encoder_rules(manifest, num_blocks=2, num_levels=4):
  """convert_ckpt_pytorch_to_tf2.py:306-372 (+ :414-416 for quant_conv)."""
  E = "first_stage_model.encoder."
  r = {"quant_conv/kernel": ("first_stage_model.quant_conv.weight", _c1),
       "quant_conv/bias": ("first_stage_model.quant_conv.bias", _ID),
       "encoder/conv_in/kernel": (E + "conv_in.weight", _conv), "encoder/conv_in/bias": (E + "conv_in.bias", _ID),
       "encoder/group_norm/gamma": (E + "norm_out.weight", _ID), "encoder/group_norm/beta": (E + "norm_out.bias", _ID),
       "encoder/conv_out/kernel": (E + "conv_out.weight", _conv), "encoder/conv_out/bias": (E + "conv_out.bias", _ID)}

  def attn(ours, theirs):
    r[f"{ours}/group_norm/gamma"] = (f"{theirs}.norm.weight", _ID)
    r[f"{ours}/group_norm/beta"] = (f"{theirs}.norm.bias", _ID)
    for o, t in (("dense_query", "q"), ("dense_key", "k"), ("dense_value", "v"), ("dense_output", "proj_out")):
      r[f"{ours}/{o}/kernel"] = (f"{theirs}.{t}.weight", _c1)
      r[f"{ours}/{o}/bias"] = (f"{theirs}.{t}.bias", _ID)

  di = 0
  for lvl in range(num_levels):
    for j in range(num_blocks):
      _res_rules(f"encoder/down/{di}/residual", f"{E}down.{lvl}.block.{j}", r, temb=False, ae=True)
      attn(f"encoder/down/{di}/attention", f"{E}down.{lvl}.attn.{j}")
      di += 1
    if lvl < num_levels - 1:
      r[f"encoder/down/{di}/conv/kernel"] = (f"{E}down.{lvl}.downsample.conv.weight", _conv)
      r[f"encoder/down/{di}/conv/bias"] = (f"{E}down.{lvl}.downsample.conv.bias", _ID)
      di += 1
  _res_rules("encoder/middle/residual1", E + "mid.block_1", r, temb=False, ae=True)
  attn("encoder/middle/attention", E + "mid.attn_1")
  _res_rules("encoder/middle/residual2", E + "mid.block_2", r, temb=False, ae=True)
  return {k: v for k, v in r.items() if k in manifest}


def _ae_manifest_rules(cfg, with_encoder, kl=True):
  cfg = dict(cfg or {})
  am = Wt.decoder_manifest(**cfg)
  nb, nl = cfg.get("num_blocks", 2), len(cfg.get("multipliers", (1, 2, 4, 4)))
  rules = decoder_rules(am, nb, nl)
  if with_encoder:
    em = Wt.encoder_manifest(**cfg, double_z=kl)
    am.update(em)
    rules.update(encoder_rules(em, nb, nl))
  return am, rules


def _apply(manifest, rules, sd, what):
  missing = [k for k in manifest if k not in rules]
  if missing:
    raise KeyError(f"{what}: no CompVis mapping for {missing[:3]} (+{len(missing) - 3 if len(missing) > 3 else 0})")
  out = {}
  for name, (shape, _) in manifest.items():
    key, fn = rules[name]
    if key not in sd:
      raise KeyError(f"{what}: checkpoint lacks {key!r} (needed for {name})")
    a = fn(sd[key]).astype(np.float32)
    if tuple(a.shape) != tuple(shape):
      raise ValueError(f"{what}: {key} -> {name}: shape {a.shape}, expected {tuple(shape)}")
    out[name] = a
  return out


def from_compvis_state_dict(sd, unet_cfg=None, transformer_cfg=None, autoencoder_cfg=None,
                            with_encoder=None, kl=True):
  """`sd`: name -> ndarray.  Returns dict(unet=..., cond_stage_model=..., autoencoder=...)
  of reference-layout float32 weights for the txt2img-f8 configuration (or the given ones).
  `with_encoder` (default: if the checkpoint has one) adds encoder/* and quant_conv/*."""
  um = Wt.unet_manifest(**(unet_cfg or {}))
  tm = Wt.transformer_manifest(**(transformer_cfg or dict(encoder_stack_size=32, hidden_size=1280, filter_size=5120)))
  if with_encoder is None:
    with_encoder = "first_stage_model.encoder.conv_in.weight" in sd
  am, arules = _ae_manifest_rules(autoencoder_cfg, with_encoder, kl)
  heads_u = (unet_cfg or {}).get("num_heads", 8)
  heads_t = (transformer_cfg or {}).get("num_heads", 8)
  return {
      "unet": _apply(um, unet_rules(um, heads_u), sd, "unet"),
      "cond_stage_model": _apply(tm, transformer_rules(tm, heads_t), sd, "cond_stage_model"),
      "autoencoder": _apply(am, arules, sd, "autoencoder"),
  }


# ---- inverse direction (export / tests) ---------------------------------------------------

def _inv(fn, a):
  a = np.asarray(a)
  if fn is _conv:
    return np.ascontiguousarray(a.transpose(3, 2, 0, 1))
  if fn is _lin:
    return np.ascontiguousarray(a.T)
  if fn is _c1:
    return np.ascontiguousarray(a.T[:, :, None, None])
  if fn is _ID:
    return np.ascontiguousarray(a)
  if a.ndim == 3 and getattr(fn, "_kind", None) == "split":
    return np.ascontiguousarray(a.reshape(a.shape[0], -1).T)
  if a.ndim == 3 and getattr(fn, "_kind", None) == "merge":
    return np.ascontiguousarray(a.reshape(-1, a.shape[2]).T)
  raise TypeError("unknown transform")


def to_compvis_state_dict(weights, unet_cfg=None, transformer_cfg=None, autoencoder_cfg=None, kl=True):
  """Inverse of from_compvis_state_dict: reference-layout weights -> CompVis names/layouts."""
  um = Wt.unet_manifest(**(unet_cfg or {}))
  tm = Wt.transformer_manifest(**(transformer_cfg or dict(encoder_stack_size=32, hidden_size=1280, filter_size=5120)))
  am, arules = _ae_manifest_rules(autoencoder_cfg, "encoder/conv_in/kernel" in weights["autoencoder"], kl)
  sd = {}
  for part, m, rules in (("unet", um, unet_rules(um, (unet_cfg or {}).get("num_heads", 8))),
                         ("cond_stage_model", tm, transformer_rules(tm, (transformer_cfg or {}).get("num_heads", 8))),
                         ("autoencoder", am, arules)):
    for name in m:
      key, fn = rules[name]
      sd[key] = _inv(fn, weights[part][name])
  return sd
