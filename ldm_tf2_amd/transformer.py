"""Text-conditioning transformer on the HIP path -- host side.

Mirrors the reference's `TransformerModel` (transformer.py:218-272): constructor
kwargs = the YAML `cond_stage_model` section; `model(token_ids int[R,77]) ->
[R,77,hidden]`.  32 pre-LN encoder layers (EncoderLayer, :173-182), no mask
(:255), exact-erf gelu, final LayerNorm (:214).

Rows of `token_ids` are usually B copies of the empty prompt followed by B
copies of the prompt (run_ldm_sampler.py:42-45); identical rows are encoded once
and copied (results are identical: no cross-row op exists in the encoder).
"""
from __future__ import annotations

import numpy as np
import torch

from . import layout as L
from . import ops
from .weights import init_weights, transformer_manifest

LAYER_NORM_EPS = 1e-5   # transformer.py:11


class _Layer:
  def __init__(self, w, p, sp, dtype, dev):
    g = lambda n: w[p + "/" + n]
    self.ln_mha = (L.vec(g("layernorm_mha/gamma"), dev), L.vec(g("layernorm_mha/beta"), dev))
    self.qk = torch.cat([L.split_kernel(g("mha/query/kernel"), sp, dtype, dev),
                         L.split_kernel(g("mha/key/kernel"), sp, dtype, dev)], 0).contiguous()
    self.v = L.split_kernel(g("mha/value/kernel"), sp, dtype, dev)
    self.o = (L.merge_kernel(g("mha/output/kernel"), sp, dtype, dev), L.vec(g("mha/output/bias"), dev))
    self.ln_ffn = (L.vec(g("layernorm_ffn/gamma"), dev), L.vec(g("layernorm_ffn/beta"), dev))
    self.f1 = (L.dense_kernel(g("ffn/filter/kernel"), dtype, dev), L.vec(g("ffn/filter/bias"), dev))
    self.f2 = (L.dense_kernel(g("ffn/output/kernel"), dtype, dev), L.vec(g("ffn/output/bias"), dev))


class TransformerModel:
  """Same kwargs as transformer.py:219-229; build-only extras: weights, dtype, device."""

  def __init__(self, vocab_size, encoder_stack_size=6, hidden_size=512, num_heads=8,
               size_per_head=64, max_seq_len=77, filter_size=2048, dropout_rate=0.1, *,
               weights=None, dtype=torch.float32, device="cuda:0", init="keras", seed=2,
               dedup_rows=True):
    self._vocab_size, self._encoder_stack_size = vocab_size, encoder_stack_size
    self._hidden_size, self._num_heads, self._size_per_head = hidden_size, num_heads, size_per_head
    self._max_seq_len, self._filter_size, self._dropout_rate = max_seq_len, filter_size, dropout_rate
    self.dtype, self.device = dtype, torch.device(device)
    self.dedup_rows = dedup_rows
    self.manifest = transformer_manifest(vocab_size, encoder_stack_size, hidden_size, num_heads,
                                         size_per_head, max_seq_len, filter_size)
    if weights is None:
      weights = init_weights(self.manifest, seed=seed, mode=init, scope="cond_stage_model")
    missing = [k for k in self.manifest if k not in weights]
    if missing:
      raise KeyError(f"TransformerModel weights missing {len(missing)} tensors, e.g. {missing[:3]}")
    self.sp = L.padded_head(size_per_head)
    dev = self.device
    self.layers = [_Layer(weights, f"encoder/layers/{i}", self.sp, dtype, dev)
                   for i in range(encoder_stack_size)]
    self.ln_out = (L.vec(weights["encoder/layernorm/gamma"], dev), L.vec(weights["encoder/layernorm/beta"], dev))
    self.tok = L.vec(weights["embedding"], dev)
    self.pos = L.vec(weights["positional_embedding"], dev)
    self.buf = L.Buffers(dev)
    self._ws = ops.new_workspace(dev)

  def _encode(self, ids):
    """ids int64 device [R,T] -> [R,T,D] (transformer.py:257-272)."""
    with ops.workspace_scope(self._ws):
      return self._encode_rows(ids)

  def _encode_rows(self, ids):
    B_, dt = self.buf, self.dtype
    R, T = ids.shape
    D, H, sp = self._hidden_size, self._num_heads, self.sp
    hs = H * sp
    scale = self._size_per_head ** -0.5
    x = B_.get("x", (R, T, D), dt)
    y = B_.get("y", (R, T, D), dt)
    ln = B_.get("ln", (R, T, D), dt)
    ops.embedding(ids, self.tok, self.pos, x)
    tp = (T + 7) // 8 * 8
    qk = B_.get("qk", (R, T, 2 * hs), dt)
    vt = B_.get("vt", (R, hs, tp), dt, zero=True)
    att = B_.get("att", (R, T, hs), dt)
    ff = B_.get("ff", (R, T, self._filter_size), dt)
    for l in self.layers:
      ops.layernorm(x, l.ln_mha[0], l.ln_mha[1], ln, LAYER_NORM_EPS)
      ops.linear(ln, l.qk, qk)
      ops.bmm_nt(ln, l.v, vt, transposed_out=True)
      ops.attention(qk[..., :hs], qk[..., hs:], vt, att, H, sp, scale)
      ops.linear(att, l.o[0], y, bias=l.o[1], residual=x)
      ops.layernorm(y, l.ln_ffn[0], l.ln_ffn[1], ln, LAYER_NORM_EPS)
      ops.linear(ln, l.f1[0], ff, bias=l.f1[1], act=ops.ACT_GELU)
      ops.linear(ff, l.f2[0], x, bias=l.f2[1], residual=y)
    out = torch.empty(R, T, D, dtype=dt, device=self.device)
    ops.layernorm(x, self.ln_out[0], self.ln_out[1], out, LAYER_NORM_EPS)
    return out

  def __call__(self, token_ids, padding_mask=None, training=False):
    if training:
      raise NotImplementedError("the HIP path is inference only")
    ids_np = np.asarray(token_ids.cpu() if isinstance(token_ids, torch.Tensor) else token_ids).astype(np.int64)
    if ids_np.ndim != 2 or ids_np.shape[1] > self._max_seq_len:
      raise ValueError(f"token_ids must be [rows, <= {self._max_seq_len}]")
    if self.dedup_rows:
      uniq, inverse = np.unique(ids_np, axis=0, return_inverse=True)
      inverse = np.asarray(inverse).reshape(-1)
    else:
      uniq, inverse = ids_np, np.arange(ids_np.shape[0])
    enc = self._encode(torch.from_numpy(np.ascontiguousarray(uniq)).to(self.device))
    if len(uniq) == ids_np.shape[0] and np.array_equal(inverse, np.arange(len(uniq))):
      return enc
    out = torch.empty(ids_np.shape[0], enc.shape[1], enc.shape[2], dtype=self.dtype, device=self.device)
    for r, u in enumerate(inverse):
      ops.cast(enc[int(u)], out[r])
    return out
