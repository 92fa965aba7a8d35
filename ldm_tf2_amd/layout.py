"""Re-layout ("compile") of reference-layout master weights into the device
layouts the HIP kernels consume, plus the scratch-buffer cache the host models use.

Reference layouts (weights.py)          -> device layouts (include/ldm_hip.h)
  conv kernel HWIO [3,3,Cin,Cout]       -> OHWI matrix [Cout, 9*Cin]   (K contiguous)
  Dense kernel [in, out]                -> [out, in]
  Projection split kernel [D, H, S]     -> [H*Sp, D], rows s >= S of each head zero
  Projection merge kernel [H, S, D]     -> [D, H*Sp], columns s >= S zero
  GEGLU Dense [C, 8C] (value | gate)    -> [8C, C] rows in blocks of 64:
                                           32 value rows then their 32 gate rows
Sp = head size padded up to the next size the attention kernel is built for
(zero padding contributes nothing to q.k^T and yields zero output columns).
This is plumbing done once at model-build time with torch tensor ops on the host;
no sampling-path arithmetic happens here.
"""
from __future__ import annotations

import numpy as np
import torch

ATTN_SP = (32, 48, 64, 80, 96, 160)   # padded head sizes instantiated in attention.hip


def padded_head(s: int) -> int:
  for sp in ATTN_SP:
    if s <= sp:
      return sp
  raise ValueError(f"head size {s} > {ATTN_SP[-1]} is not supported by the fused attention kernel")


def _dev(a, dtype, device):
  t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
  return t.to(dtype).contiguous().to(device)


def conv_kernel(k_hwio, dtype, device):
  k = torch.from_numpy(np.ascontiguousarray(k_hwio))
  cout = k.shape[3]
  return _dev(k.permute(3, 0, 1, 2).reshape(cout, -1), dtype, device)


def conv_shortcut_kernel(k_hwio, k_io, dtype, device):
  """[Cout, 9*Cin + Cin2]: the 3x3 kernel's OHWI rows followed by the 1x1 shortcut's [out, in] rows -- ldm_gemm's
  K order with a second A operand (conv3x3(..., x2=...): the ResBlock shortcut inside its second convolution)."""
  k = torch.from_numpy(np.ascontiguousarray(k_hwio))
  cout = k.shape[3]
  sc = torch.from_numpy(np.ascontiguousarray(k_io)).t()
  assert sc.shape[0] == cout
  return _dev(torch.cat([k.permute(3, 0, 1, 2).reshape(cout, -1), sc], 1), dtype, device)


def dense_kernel(k_io, dtype, device):
  return _dev(torch.from_numpy(np.ascontiguousarray(k_io)).t(), dtype, device)


def split_kernel(k_dhs, sp, dtype, device):
  k = torch.from_numpy(np.ascontiguousarray(k_dhs))          # [D, H, S]
  d, h, s = k.shape
  out = torch.zeros(h, sp, d, dtype=k.dtype)
  out[:, :s, :] = k.permute(1, 2, 0)
  return _dev(out.reshape(h * sp, d), dtype, device)


def merge_kernel(k_hsd, sp, dtype, device):
  k = torch.from_numpy(np.ascontiguousarray(k_hsd))          # [H, S, D]
  h, s, d = k.shape
  out = torch.zeros(d, h, sp, dtype=k.dtype)
  out[:, :, :s] = k.permute(2, 0, 1)
  return _dev(out.reshape(d, h * sp), dtype, device)


def geglu_kernel(k_io, bias, dtype, device):
  k = torch.from_numpy(np.ascontiguousarray(k_io))           # [C, 8C]
  c, n = k.shape
  half = n // 2
  assert half % 32 == 0, "GEGLU width must be a multiple of 32"
  wt = k.t()                                                  # [8C, C]; rows: value | gate
  val, gate = wt[:half].reshape(half // 32, 32, c), wt[half:].reshape(half // 32, 32, c)
  inter = torch.stack([val, gate], dim=1).reshape(n, c)
  b = torch.from_numpy(np.ascontiguousarray(bias))
  bi = torch.stack([b[:half].reshape(-1, 32), b[half:].reshape(-1, 32)], dim=1).reshape(n)
  return _dev(inter, dtype, device), _dev(bi, torch.float32, device)


MS_DIM = 40        # the padded head dim ldm_attention_ms uses (heads of 40 padded to 48)
MS_LOG2E = 1.4426950408889634


def ms_ones(heads, sp, offset=0, total=None):
  """float32 vector with 1.0 at dim MS_DIM of every head (rows offset + h*sp + MS_DIM), else 0."""
  v = torch.zeros(total if total is not None else offset + heads * sp, dtype=torch.float32)
  v[offset + MS_DIM: offset + heads * sp: sp] = 1.0
  return v


def ln_fold(wt_nk, gamma, beta, bias, dtype, device, row_scale=None, bias_extra=None):
  """LayerNorm -> Dense folded for ldm_gemm's `ln_cs` form (include/ldm_hip.h): from the device-layout
  float32 matrix wt [N, K] (rows = outputs), the LayerNorm's gamma / beta [K] and the Dense bias [N]
  (or None) returns (w', cs, b') on the device with
      w' = dtype(gamma (.) W),  cs[n] = sum_k float(w'[n, k]),  b' = bias + W beta   (float32),
  so that LN(x) W^T + bias = rstd (x w'^T - mean cs) + b'.  cs is summed from the ROUNDED w' (what
  the MFMA multiplies), in float64, so that the mean term cancels exactly what the product carries.
  `row_scale` [N]: every output n (weights and bias) is multiplied by row_scale[n] first (the attention
  scale folded into the query projection); `bias_extra` [N] is added to b' last (the 1.0 of the padded
  head rows that ldm_attention_ms uses)."""
  w = wt_nk.detach().to("cpu", torch.float32)
  if row_scale is not None:
    rs = torch.as_tensor(np.asarray(row_scale), dtype=torch.float32)
    w = w * rs[:, None]
    if bias is not None:
      bias = np.asarray(bias, dtype=np.float32) * rs.numpy()
  g = torch.as_tensor(np.asarray(gamma), dtype=torch.float32)
  b = torch.as_tensor(np.asarray(beta), dtype=torch.float32)
  wq = (w * g[None, :]).to(dtype)
  cs = wq.to(torch.float64).sum(1).to(torch.float32)
  bb = (w.to(torch.float64) @ b.to(torch.float64)).to(torch.float32)
  if bias is not None:
    bb = bb + torch.as_tensor(np.asarray(bias), dtype=torch.float32)
  if bias_extra is not None:
    bb = bb + torch.as_tensor(np.asarray(bias_extra), dtype=torch.float32)
  return wq.contiguous().to(device), cs.contiguous().to(device), bb.contiguous().to(device)


def ff_proj_fold(ff_k_io, ff_bias, proj_k_io, proj_bias, dtype, device):
  """The feed-forward's output Dense and the SpatialTransformer's proj_out as ONE product (unet.py:313, :338, :363-365):
      out = x + Wp (h + W2 g + b2) + bp  =  x + (Wp W2) g + Wp h + (Wp b2 + bp)
  -- two linear layers with only a residual between them.  Returns ([C, 4C + C] = (Wp W2 | Wp), rows = outputs, K
  contiguous; folded bias [C]) for `ops.linear(g, w, out, bias=b, residual=x, x2=h)`.  The product Wp W2 is formed in
  float64 from the float32 master weights and rounded once."""
  w2 = torch.from_numpy(np.ascontiguousarray(ff_k_io)).to(torch.float64)       # [4C, C]  (in, out)
  wp = torch.from_numpy(np.ascontiguousarray(proj_k_io)).to(torch.float64)     # [C, C]   (in, out)
  b2 = torch.from_numpy(np.ascontiguousarray(ff_bias)).to(torch.float64)
  bp = torch.from_numpy(np.ascontiguousarray(proj_bias)).to(torch.float64)
  w = torch.cat([(w2 @ wp).t(), wp.t()], 1)                                     # [C, 4C + C]
  b = b2 @ wp + bp
  return _dev(w.to(torch.float32), dtype, device), _dev(b.to(torch.float32), torch.float32, device)


def ffn_aux(cs, bias):
  """ldm_ffn_geglu's `aux`: per 128 rows of the folded GEGLU weights their column sums (128) then their
  folded bias (128), float32 [8C / 128, 256]."""
  n = cs.numel()
  assert n % 128 == 0 and bias.numel() == n
  return torch.cat([cs.reshape(n // 128, 128), bias.reshape(n // 128, 128)], 1).contiguous()


def vec(a, device):
  return _dev(a, torch.float32, device)


class Buffers:
  """Scratch tensors keyed by (tag, shape, dtype); allocated once, reused by every
  later call with the same key.  All kernels of one model run on one stream in
  program order, so a tag only needs to be unique among tensors that are live at
  the same time."""

  def __init__(self, device):
    self.device = device
    self._b = {}

  def get(self, tag, shape, dtype, zero=False):
    key = (tag, tuple(int(s) for s in shape), dtype)
    t = self._b.get(key)
    if t is None:
      t = (torch.zeros if zero else torch.empty)(key[1], dtype=dtype, device=self.device)
      self._b[key] = t
    return t

  def nbytes(self):
    return sum(t.numel() * t.element_size() for t in self._b.values())
