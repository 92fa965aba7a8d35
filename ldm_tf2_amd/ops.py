"""Thin tensor-level wrappers over the C ABI (include/ldm_hip.h).

torch is used here only as the device-memory container and stream provider:
every function turns tensors into (pointer, stride, size) arguments and enqueues
hand-written HIP kernels from libldm_hip.so on torch's current stream.  Nothing
in this module computes with torch ops.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

from . import _lib
from ._lib import ACT_GEGLU, ACT_GELU, ACT_NONE, ACT_SILU, BF16, F32, GemmParams, check, lib

__all__ = [
    "ACT_NONE", "ACT_GELU", "ACT_GEGLU", "ACT_SILU", "F32", "BF16", "code", "linear", "conv3x3",
    "bmm_nt", "conv3x3_small", "groupnorm", "layernorm", "softmax_rows", "attention",
    "time_embedding", "gemv", "cfg_ddim_update", "post_quant", "vq_nearest", "embedding",
    "minmax_u8", "cast",
]

_TORCH_DT = {torch.float32: F32, torch.bfloat16: BF16}
_byref = C.byref      # (functions below use C for a channel count)
_WS = {}
_WS_BYTES = 96 << 20


def code(dtype) -> int:
  try:
    return _TORCH_DT[dtype]
  except KeyError:
    raise TypeError(f"unsupported dtype {dtype}: the HIP path stores float32 or bfloat16")


def _stream():
  return torch.cuda.current_stream().cuda_stream


def _ptr(t):
  if t is None:
    return None
  if not t.is_cuda:
    raise ValueError("the HIP path needs device tensors (there is no CPU fallback)")
  return t.data_ptr()


def _f32(t, what):
  if t is not None and t.dtype != torch.float32:
    raise TypeError(f"{what} must be float32")
  return t


def row_ld(t) -> int:
  """Row stride (elements) of `t` seen as a 2-D [rows, cols] matrix; the leading
  dims must collapse onto a single uniform stride (true for channel slices of
  NHWC buffers)."""
  if t.dim() < 2:
    return t.shape[-1]
  if t.shape[-1] != 1 and t.stride(-1) != 1:
    raise ValueError("last dim must be contiguous")
  ld = t.stride(-2)
  for i in range(t.dim() - 2, 0, -1):
    if t.shape[i - 1] != 1 and t.stride(i - 1) != t.stride(i) * t.shape[i]:
      raise ValueError(f"tensor with shape {tuple(t.shape)} strides {t.stride()} has no uniform row stride")
  return ld


def workspace(device):
  """Split-K partial-sum workspace for a launch made now.  ldm_gemm's slabs live from the GEMM
  launch to its reduce launch, in stream order, so one workspace serves one stream at a time:
  every model owns its own (`workspace_scope`, allocated at model build, before any graph
  capture); launches outside a model use one per (device, current stream)."""
  stack = _ws_stack()
  if stack:
    return stack[-1]
  key = (device, torch.cuda.current_stream(device).cuda_stream)
  ws = _WS.get(key)
  if ws is None:
    ws = new_workspace(device)
    _WS[key] = ws
  return ws


def new_workspace(device):
  return torch.empty(_WS_BYTES, dtype=torch.uint8, device=device)


# The scope stacks (workspace, active plan table) are per host THREAD: two samplers driven from two
# Python threads, one model each, must not see each other's workspace or plan table (the registry of
# loaded tables, _TABLES, is shared and only read inside a forward).
_TLS = threading.local()


def _ws_stack():
  st = getattr(_TLS, "ws", None)
  if st is None:
    st = _TLS.ws = []
  return st


class workspace_scope:
  """`with workspace_scope(ws): ...` -- ldm_gemm launches made by THIS thread inside use `ws` (a
  model's own workspace), so two models replaying on different streams -- from one host thread or
  from two -- never share split-K slabs."""

  def __init__(self, ws):
    self._ws = ws

  def __enter__(self):
    _ws_stack().append(self._ws)
    return self

  def __exit__(self, *exc):
    _ws_stack().pop()
    return False


# ---- per-shape launch plans -----------------------------------------------------------------
# ldm_gemm picks its tile and split-K with a cost model calibrated on isolated launches.  In
# the real step (weights cold in HBM, activations warm in L2/Infinity Cache, neighbours
# competing) the optimum differs on some shapes, so measured overrides can be loaded:
# {problem key: [tile, split_k]} JSON files written by tools/tune_step_plans.py, which times
# WHOLE captured U-Net steps while varying one shape's plan at a time.
#
# A table belongs to ONE step configuration (its file header: "config": {"rows": R, "latent":
# h, "dtype": "bf16"|"f32"}) because a problem key names a shape, not the step it sits in: the
# same shape occurs in different configurations with different in-situ optima.  Exactly one
# table is active at a time; the U-Net activates the table of its (rows, latent, dtype) for the
# duration of a forward (`plan_scope`), everything else runs on the cost model.  LDM_NO_PLANS=1
# disables the packaged tables, LDM_GEMM_PLANS=<file> registers another.  Plans only reorder
# float additions.
_TABLES = {}          # (rows, latent, dtype code) -> {key: (tile, split_k)}
_PLAN_RECORD = None


def _active():
  """This thread's active table (possibly edited in place by tools/tune_step_plans.py)."""
  return getattr(_TLS, "active", None) or {}


def _active_cfg():
  return getattr(_TLS, "active_cfg", None)
_TILE_DIMS = {1: (256, 128, 1), 2: (128, 128, 2), 3: (128, 64, 3), 4: (64, 64, 5), 6: (128, 160, 2),
              7: (256, 160, 1), 8: (128, 320, 1),       # BM, BN, resident workgroups per CU
              9: (256, 160, 1), 10: (128, 160, 2), 11: (256, 128, 1), 12: (128, 128, 2),   # bf16 16x16x32 MFMA path
              13: (256, 160, 1), 14: (256, 128, 1),     # persistent ping-pong kernel (no split-K, N % BN == 0)
              15: (256, 160, 1), 16: (256, 128, 1),     # tiles 9 / 11 with the halo-staged A operand (stride-1 convs)
              17: (64, 64, 2), 18: (128, 64, 2), 19: (128, 128, 1)}   # tiles 4 / 3 / 2 with a 4- / 3- / 3-stage LDS ring
_BF16_TILES = (9, 10, 11, 12, 13, 14, 15, 16)
_PERSISTENT_TILES = (13, 14)
_HALO_RING_TILES = (15, 16)


def _halo_ring_key(key, M, N, bn):
  """True if the problem `key` names can run on a halo-staged conv tile (gemm.hip halo_ring_ok)."""
  import re
  m = re.search(r"conv1 H(\d+) W(\d+) s1 u0 nlp0", key or "")
  if not m or " x2" in (key or ""):          # (a second A operand runs on the implicit-GEMM tiles only)
    return False
  H, W = int(m.group(1)), int(m.group(2))
  return W in (16, 32) and H % (256 // W) == 0 and M % 256 == 0 and N % bn == 0


def plan_key(p) -> str:
  key = "M%d N%d K%d b%d conv%d H%d W%d s%d u%d nlp%d act%d dt%d odt%d" % (
      p.M, p.N, p.K, p.batch, p.conv, p.H, p.W, p.stride, p.upsample, p.no_lead_pad, p.act, p.dtype, p.out_dtype)
  if p.a2:
    key += " x2"            # second A operand (the ResBlock shortcut inside the convolution): K = 9 Cin + Cin2
  if p.out2 and p.n_split == 0:
    key += " t1"            # whole product stored transposed (linear_t)
  elif p.out2 and p.ln_cs:
    key += " sp%d" % p.n_split   # q | k row-major + V^T transposed in one LayerNorm-folded launch
  if p.ln_cs:
    key += " ln1"           # LayerNorm of the rows folded in (persistent tiles only)
  return key


def _cfg_key(rows, latent, dtype):
  dt = dtype if isinstance(dtype, int) else (BF16 if str(dtype) in ("bf16", "torch.bfloat16") else F32)
  return (int(rows), int(latent), dt)


def load_plans(path):
  """Registers the {key: [tile, split_k]} table of one JSON file under its header's
  configuration; returns that configuration key.  Two files for one configuration that
  disagree on a key are an error (a silently merged table is not the one that was measured)."""
  import json
  with open(path) as f:
    d = json.load(f)
  c = d.get("config")
  if not c:
    raise ValueError(f"{path}: plan table without a \"config\" header (rows / latent / dtype)")
  ck = _cfg_key(c["rows"], c["latent"], c["dtype"])
  table = _TABLES.setdefault(ck, {})
  for k, v in d["plans"].items():
    plan = (int(v[0]), int(v[1]))
    if table.get(k, plan) != plan:
      raise ValueError(f"{path}: plan for '{k}' conflicts with an already loaded table of config {ck}")
    table[k] = plan
  return ck


def select_plans(rows=None, latent=None, dtype=None):
  """Makes the table of one step configuration the active one (none when no table is
  registered for it, or when called without arguments).  Returns the previous selection."""
  prev = _active_cfg()
  ck = None if rows is None else _cfg_key(rows, latent, dtype)
  _TLS.active_cfg = ck
  _TLS.active = _TABLES.get(ck, {}) if ck is not None else {}
  return prev


class plan_scope:
  """`with plan_scope(rows, latent, dtype): ...` -- the launches inside use that table."""

  def __init__(self, rows, latent, dtype):
    self._cfg = (rows, latent, dtype)

  def __enter__(self):
    self._prev = select_plans(*self._cfg)
    return self

  def __exit__(self, *exc):
    if self._prev is None:
      select_plans()
    else:
      select_plans(*self._prev)
    return False


def set_plan(key, plan):
  """Edits the ACTIVE table: plan = (tile, split_k) or None to drop the override
  (tools/tune_step_plans.py)."""
  cfg = _active_cfg()
  if cfg is None:
    raise RuntimeError("set_plan: no configuration selected (select_plans first)")
  table = _TABLES.setdefault(cfg, {})
  if plan is None:
    table.pop(key, None)
  else:
    table[key] = (int(plan[0]), int(plan[1]))
  select_plans(*cfg)


def gemm_plans(rows=None, latent=None, dtype=None):
  """Copy of one configuration's table (the active one by default)."""
  if rows is None:
    return dict(_active())
  return dict(_TABLES.get(_cfg_key(rows, latent, dtype), {}))


def plan_tables():
  return {k: dict(v) for k, v in _TABLES.items()}


def clear_plans():
  _TABLES.clear()
  select_plans()


def record_plan_keys(sink):
  """While `sink` is a dict, every auto-planned ldm_gemm launch stores key -> (M, N, K, batch,
  act, dtype) in it (tools/tune_step_plans.py enumerates a step's problems this way)."""
  global _PLAN_RECORD
  _PLAN_RECORD = sink


def plan_candidates(M, N, K, batch, act, dtype, key=None):
  """(tile, split_k) pairs worth timing for one problem (`key`: its plan key, which carries the conv
  geometry the halo-staged tiles depend on)."""
  ktiles = K // (64 if dtype == BF16 else 32)
  out = []
  for tile, (bm, bn, res) in _TILE_DIMS.items():
    if tile in _HALO_RING_TILES and not (dtype == BF16 and batch == 1 and _halo_ring_key(key, M, N, bn)):
      continue
    if act == ACT_GEGLU and tile not in (1, 2, 11, 12, 14, 19):
      continue
    if tile in _BF16_TILES and dtype != BF16:
      continue
    if bn % 160 == 0 and N % bn != 0:
      continue
    tiles = -(-M // bm) * -(-N // bn) * batch
    if tile in _PERSISTENT_TILES:
      if N % bn == 0 and batch == 1 and tiles >= 128 and " x2" not in (key or ""):
        out.append((tile, 1))
      continue
    out.append((tile, 1))
    if batch == 1 and tiles < 256 * res:         # sub-round launches: split-K candidates
      out += [(tile, s) for s in (2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24)
              if ktiles // s >= 4 and tiles * s <= 512 * res]
  return out


def _load_default_plans():
  d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plans")
  if os.path.isdir(d) and os.environ.get("LDM_NO_PLANS") is None:
    for name in sorted(os.listdir(d)):
      if name.endswith(".json"):
        load_plans(os.path.join(d, name))
  extra = os.environ.get("LDM_GEMM_PLANS")
  if extra:
    load_plans(extra)


_load_default_plans()


def resolve_plan(p: GemmParams):
  """Applies the active plan table to an auto-planned problem (tile == 0 and split_k == 0); True if
  the table supplied the plan."""
  active = _active()
  if p.tile == 0 and p.split_k == 0 and not p.ln_out and (active or _PLAN_RECORD is not None):
    key = plan_key(p)
    if _PLAN_RECORD is not None:
      _PLAN_RECORD[key] = (p.M, p.N, p.K, p.batch, p.act, p.dtype)
    plan = active.get(key)
    if plan is None and p.ln_cs:
      # a LayerNorm-fold launch without an entry of its own inherits the plain product's plan when
      # that names a persistent tile (13 = 160-column, 14 = 128-column n-tiles)
      plan = active.get(key[:-len(" ln1")])
      if plan is not None and plan[0] not in _PERSISTENT_TILES:
        plan = None
    if plan is not None:
      p.tile, p.split_k = plan
      return True
  return False


class PendingReduce:
  """A split-K product whose float32 slabs wait in the workspace (ldm_gemm with defer_reduce): complete it
  with `finish` (plain reduce) or `groupnorm(..., pending=...)` (reduce + GroupNorm in one launch) before
  anything else uses that workspace or reads `out`."""

  def __init__(self, p, out, ws, keep):
    self.p, self.out, self.ws, self.keep, self.done = p, out, ws, keep, False


def _outstanding():
  st = getattr(_TLS, "pending", None)
  if st is None:
    st = _TLS.pending = {}
  return st


def finish(pending):
  """Completes a deferred split-K product with the plain reduce + epilogue launch."""
  if pending is None or pending.done:
    return
  check(lib.ldm_gemm_reduce(C.byref(pending.p), _stream()), "ldm_gemm_reduce")
  pending.done = True
  _outstanding().pop(pending.ws.data_ptr(), None)


def _gemm_deferred(p: GemmParams, device, out, keep):
  """ldm_gemm with the reduce left to the caller.  Returns a PendingReduce, or None when the plan does
  not split K (the product is then complete)."""
  ws = workspace(device)
  if ws.data_ptr() in _outstanding():
    raise RuntimeError("ldm_gemm: this workspace still holds the slabs of a deferred split-K product")
  p.workspace = ws.data_ptr()
  p.workspace_bytes = ws.numel()
  planned = resolve_plan(p)
  while True:
    split = lib.ldm_gemm_splits(C.byref(p)) > 1
    p.defer_reduce = 1 if split else 0
    st = lib.ldm_gemm(C.byref(p), _stream())
    if st in (_lib.ERR_ARG, _lib.ERR_WORKSPACE) and planned:
      # the table's tile cannot run THIS launch (a plan key names the shape, not the epilogue; see _gemm): once
      # more on the cost model, without consulting the table again.  Nothing was enqueued by the rejected call.
      planned = False
      p.tile, p.split_k = 0, 0
      continue
    check(st, "ldm_gemm")
    break
  if not split:
    return None
  pend = PendingReduce(p, out, ws, keep)
  _outstanding()[ws.data_ptr()] = pend
  return pend


def _gemm(p: GemmParams, device):
  """Every ldm_gemm launch of the package goes through here (tools/gemm_hooks.py wraps this
  one function for measurements; the product path itself carries no hooks)."""
  ws = workspace(device)
  if ws.data_ptr() in _outstanding():
    raise RuntimeError("ldm_gemm: this workspace still holds the slabs of a deferred split-K product "
                       "(ops.finish it, or let the GroupNorm that follows consume it, first)")
  p.workspace = ws.data_ptr()
  p.workspace_bytes = ws.numel()
  planned = resolve_plan(p)
  st = lib.ldm_gemm(C.byref(p), _stream())
  if st in (_lib.ERR_ARG, _lib.ERR_WORKSPACE) and planned:
    # A plan key names the shape, not the epilogue: the table's persistent / halo tile exists only for
    # the epilogue variants that are instantiated.  A launch that shares a key with a tuned one but not
    # its epilogue (another U-Net configuration) runs on the cost model instead of failing.  Nothing
    # was enqueued by the rejected call (argument checks precede every launch).
    p.tile, p.split_k = 0, 0
    st = lib.ldm_gemm(C.byref(p), _stream())
  check(st, "ldm_gemm")


def linear_ln_supported(n_out, dtype):
  """True if `linear(..., ln=...)` can emit the LayerNorm of its output rows in the same launch."""
  return bool(lib.ldm_gemm_ln_supported(int(n_out), code(dtype)))


def linear(x, wt, out, bias=None, act=ACT_NONE, residual=None, addend=None, add_rows=0,
           alpha=1.0, tile=0, split_k=0, ln=None, out2=None, ln_fold=None, x2=None, defer_reduce=False):
  """out[..., n] = act(alpha * x[..., :] . wt[n, :] + bias[n] + addend[group]) + residual.
  x [..., K]; wt [N, K] contiguous; out [..., N] (N/2 wide for GEGLU).
  `ln=(gamma, beta, ln_out, eps)`: also writes ln_out = LayerNorm(out) (same shape/dtype).
  `out2` [G, N2, T']: the weight rows beyond out's width are a second projection whose result is
  stored TRANSPOSED per group of T = M/G rows (q|k into `out`, v into the attention kernel's
  V^T [rows, heads*Sp, T] in one launch).
  `x2` [..., K2]: out = x . wt[:, :K]^T + x2 . wt[:, K:]^T + ...: two products over the same rows as ONE launch.
  `defer_reduce`: as conv3x3's -- a split-K plan leaves its slabs to the caller (returns a PendingReduce or None).
  `ln_fold=(cs, eps)`: out = LayerNorm(x) . W^T + b with the normalisation folded into the product:
  `wt` holds gamma (.) W, `bias` holds b + W beta, cs[n] = sum_k wt[n, k] (layout.ln_fold); the kernel
  derives the row statistics itself (bf16, persistent tiles)."""
  K1 = x.shape[-1]
  K2 = 0 if x2 is None else x2.shape[-1]
  K = K1 + K2
  N = wt.shape[0]
  M = x.numel() // K1
  assert wt.shape[1] == K and wt.is_contiguous() and wt.dtype == x.dtype
  p = GemmParams()
  p.a, p.w, p.out = _ptr(x), _ptr(wt), _ptr(out)
  p.bias = _ptr(_f32(bias, "bias"))
  p.addend = _ptr(_f32(addend, "addend"))
  p.residual = _ptr(residual)
  if residual is not None:
    assert residual.dtype == out.dtype
    p.ldr = row_ld(residual)
  p.lda, p.ldc_m, p.ldc_n = row_ld(x), row_ld(out), 1
  p.M, p.N, p.K, p.batch = M, N, K, 1
  if x2 is not None:     # the last K2 columns of K multiply rows of a second matrix (ldm_gemm a2)
    assert x2.numel() // K2 == M and x2.dtype == x.dtype
    p.a2, p.lda2, p.Cin2 = _ptr(x2), row_ld(x2), K2
  if addend is not None:
    p.add_rows = add_rows if add_rows > 0 else M
    p.add_ld = addend.stride(0) if (addend.dim() == 2 and addend.shape[0] > 1) else 0
  p.act, p.dtype, p.out_dtype, p.alpha = act, code(x.dtype), code(out.dtype), alpha
  p.tile, p.split_k = tile, split_k
  if out2 is not None:
    assert out2.dtype == out.dtype and out2.dim() == 3 and out2.stride(2) == 1 and M % out2.shape[0] == 0
    assert out.shape[-1] + out2.shape[1] == N and out2.shape[2] >= M // out2.shape[0]
    p.out2, p.n_split, p.rows2 = _ptr(out2), out.shape[-1], M // out2.shape[0]
    p.ld2, p.stride2 = out2.stride(1), out2.stride(0)
  if ln is not None:
    gamma, beta, ln_out, eps = ln
    assert ln_out.dtype == out.dtype and tuple(ln_out.shape) == tuple(out.shape)
    p.ln_out, p.ld_ln, p.ln_eps = _ptr(ln_out), row_ld(ln_out), float(eps)
    p.ln_gamma, p.ln_beta = _ptr(_f32(gamma, "ln gamma")), _ptr(_f32(beta, "ln beta"))
  if ln_fold is not None:
    cs, eps = ln_fold
    assert bias is not None and tuple(cs.shape) == (N,) and cs.is_contiguous()
    p.ln_cs, p.ln_eps = _ptr(_f32(cs, "ln_fold column sums")), float(eps)
  if defer_reduce:
    # -> PendingReduce when the plan splits K (the caller owes `finish` or a consuming groupnorm), else None
    return _gemm_deferred(p, x.device, out, (x, wt, bias, addend, residual, x2))
  _gemm(p, x.device)
  return out


def linear_t_supported(x, wt, out_t):
  """True if `linear_t` can run this product (persistent kernel: bf16, N a multiple of 128 or 160,
  groups of rows that are multiples of 32)."""
  K, N = x.shape[-1], wt.shape[0]
  M = x.numel() // K
  G = out_t.shape[0]
  return (x.dtype == torch.bfloat16 and out_t.dtype == torch.bfloat16 and K % 64 == 0 and
          (N % 128 == 0 or N % 160 == 0) and M % G == 0 and (M // G) % 32 == 0 and out_t.stride(2) == 1 and
          out_t.stride(1) % 8 == 0 and out_t.stride(0) % 8 == 0)


def linear_t(x, wt, out_t, tile=0, bias=None, ln_fold=None):
  """out_t[g, n, t] = sum_k x[g * T + t, k] * wt[n, k] with T = M / G: the product stored
  TRANSPOSED per group of T rows (the self-attention V projection lands directly in the attention
  kernel's V^T [rows, heads*Sp, T]).  Runs on the persistent kernel (tile 13 / 14; the plan table
  may name which)."""
  K = x.shape[-1]
  N = wt.shape[0]
  M = x.numel() // K
  G = out_t.shape[0]
  assert wt.shape[1] == K and wt.is_contiguous() and wt.dtype == x.dtype
  assert out_t.dim() == 3 and out_t.shape[1] == N and out_t.shape[2] >= M // G
  p = GemmParams()
  p.a, p.w = _ptr(x), _ptr(wt)
  p.out = p.out2 = _ptr(out_t)
  p.lda, p.ldc_m, p.ldc_n = row_ld(x), N, 1
  p.M, p.N, p.K, p.batch = M, N, K, 1
  p.act, p.dtype, p.out_dtype, p.alpha = ACT_NONE, code(x.dtype), code(out_t.dtype), 1.0
  p.n_split, p.rows2, p.ld2, p.stride2 = 0, M // G, out_t.stride(1), out_t.stride(0)
  p.tile = tile
  if ln_fold is not None:                # LayerNorm of the rows folded in (see linear)
    cs, eps = ln_fold
    assert bias is not None and tuple(cs.shape) == (N,) and cs.is_contiguous()
    p.bias = _ptr(_f32(bias, "bias"))
    p.ln_cs, p.ln_eps = _ptr(_f32(cs, "ln_fold column sums")), float(eps)
  resolve_plan(p)
  if p.tile not in _PERSISTENT_TILES:
    p.tile, p.split_k = (14 if N % 128 == 0 else 13), 0
  _gemm(p, x.device)
  return out_t


def _conv_params(x, wt, out, bias, stride, upsample, addend, residual, tile, split_k, no_lead_pad=False, x2=None):
  B, H, W, Cin = x.shape
  Cout = wt.shape[0]
  hs, ws_ = (2 * H, 2 * W) if upsample else (H, W)
  pads = 1 if no_lead_pad else 2
  OH, OW = (hs + pads - 3) // stride + 1, (ws_ + pads - 3) // stride + 1
  Cin2 = 0 if x2 is None else x2.shape[-1]
  assert wt.shape[1] == 9 * Cin + Cin2 and wt.is_contiguous() and wt.dtype == x.dtype
  assert tuple(out.shape) == (B, OH, OW, Cout), (tuple(out.shape), (B, OH, OW, Cout))
  p = GemmParams()
  p.a, p.w, p.out = _ptr(x), _ptr(wt), _ptr(out)
  p.bias = _ptr(_f32(bias, "bias"))
  p.addend = _ptr(_f32(addend, "addend"))
  p.residual = _ptr(residual)
  if residual is not None:
    assert residual.dtype == out.dtype
    p.ldr = row_ld(residual)
  p.lda, p.ldc_m, p.ldc_n = row_ld(x), row_ld(out), 1
  p.M, p.N, p.K, p.batch = B * OH * OW, Cout, 9 * Cin + Cin2, 1
  if x2 is not None:
    assert tuple(x2.shape[:3]) == (B, OH, OW) and x2.dtype == x.dtype and stride == 1 and not upsample
    p.a2, p.lda2, p.Cin2 = _ptr(x2), row_ld(x2), Cin2
  if addend is not None:
    p.add_rows = OH * OW
    p.add_ld = addend.stride(0) if (addend.dim() == 2 and addend.shape[0] > 1) else 0
  p.conv, p.B, p.H, p.W, p.Cin, p.OH, p.OW = 1, B, H, W, Cin, OH, OW
  p.stride, p.upsample, p.no_lead_pad = stride, int(bool(upsample)), int(bool(no_lead_pad))
  p.act, p.dtype, p.out_dtype, p.alpha = ACT_NONE, code(x.dtype), code(out.dtype), 1.0
  p.tile, p.split_k = tile, split_k
  return p


def conv3x3(x, wt, out, bias=None, stride=1, upsample=False, addend=None, residual=None,
            tile=0, split_k=0, no_lead_pad=False, defer_reduce=False, x2=None):
  """3x3 convolution, NHWC, pad 1 (Keras SAME for stride 1; the U-Net's explicit
  pad(1,1)+VALID for stride 2; `no_lead_pad`: the autoencoder's pad (0,1),(0,1)+VALID stride-2
  downsample, autoencoder.py:133), optional fused nearest-2x upsample of the input.
  x [B,H,W,Cin] (channel slice allowed); wt [Cout, 9*Cin] = OHWI; out [B,OH,OW,Cout].
  `x2` [B,H,W,Cin2] (stride 1): out += x2 . wt[:, 9*Cin:]^T at the output pixel -- the ResBlock's 1x1 shortcut
  (unet.py:393-397) inside this convolution's K loop; wt is then [Cout, 9*Cin + Cin2] (layout.conv_shortcut_kernel)."""
  p = _conv_params(x, wt, out, bias, stride, upsample, addend, residual, tile, split_k, no_lead_pad, x2)
  if defer_reduce:
    # -> PendingReduce when the plan splits K (the caller owes `finish` or a consuming groupnorm), else None
    return _gemm_deferred(p, x.device, out, (x, wt, bias, addend, residual))
  _gemm(p, x.device)
  return out


def bmm_nt(a, w, out, alpha=1.0, bias=None, transposed_out=False, tile=0):
  """Batched out[b] = alpha * a[b] @ w[b]^T (+bias).  a [Bt, M, K]; w [Bt, N, K] or
  [N, K] (shared); out [Bt, M, N], or [Bt, N, ldn>=M] when transposed_out."""
  Bt, M, K = a.shape
  shared = w.dim() == 2
  N = w.shape[-2]
  assert a.stride(-1) == 1 and w.stride(-1) == 1 and out.stride(-1) == 1
  assert w.stride(-2) == K, "weights / second operand rows must be dense"
  p = GemmParams()
  p.a, p.w, p.out = _ptr(a), _ptr(w), _ptr(out)
  p.bias = _ptr(_f32(bias, "bias"))
  p.lda = a.stride(1)
  p.stride_a = a.stride(0)
  p.stride_w = 0 if shared else w.stride(0)
  p.stride_c = out.stride(0)
  if transposed_out:
    p.ldc_m, p.ldc_n = 1, out.stride(1)
  else:
    p.ldc_m, p.ldc_n = out.stride(1), 1
  p.M, p.N, p.K, p.batch = M, N, K, Bt
  p.act, p.dtype, p.out_dtype, p.alpha = ACT_NONE, code(a.dtype), code(out.dtype), alpha
  p.tile = tile
  _gemm(p, a.device)
  return out


def conv3x3_small(x, kernel_hwio, bias, out):
  B, H, W, Cin = x.shape
  Cout = kernel_hwio.shape[-1]
  assert kernel_hwio.is_contiguous() and kernel_hwio.dtype == torch.float32
  check(lib.ldm_conv3x3_small(_ptr(x), row_ld(x), code(x.dtype), _ptr(kernel_hwio),
                              _ptr(_f32(bias, "bias")), _ptr(out), row_ld(out), code(out.dtype),
                              B, H, W, Cin, Cout, _stream()), "ldm_conv3x3_small")
  return out


def groupnorm(x, gamma, beta, out, eps, silu=False, groups=32, partial=None, fused=None, pending=None,
              store_x=True):
  """x, out [B, H, W, C] (channel slices allowed).  Small images take the single-launch
  kernel (`fused`; default: whenever the library supports the shape), large ones the
  partial-sums + apply pair.
  `pending`: x is the output of a deferred split-K product (conv3x3(..., defer_reduce=True)): its reduce +
  epilogue and this GroupNorm run as ONE launch (ldm_groupnorm_splitk); x itself is written only when
  `store_x` (a ResBlock's conv1 output has no other reader)."""
  B, C = x.shape[0], x.shape[-1]
  HW = x.numel() // (B * C)
  if pending is not None and not pending.done:
    same = (pending.out.data_ptr() == x.data_ptr() and tuple(pending.out.shape) == tuple(x.shape) and
            pending.out.stride() == x.stride())
    if same and lib.ldm_groupnorm_splitk_supported(B, HW, C, groups, code(x.dtype)) and row_ld(x) == pending.p.ldc_m:
      assert out.dtype == x.dtype
      check(lib.ldm_groupnorm_splitk(_byref(pending.p), _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")),
                                     _ptr(out), row_ld(out), B, HW, groups, float(eps), int(bool(silu)),
                                     int(bool(store_x)), _stream()), "ldm_groupnorm_splitk")
      pending.done = True
      _outstanding().pop(pending.ws.data_ptr(), None)
      return out
    finish(pending)
  if fused is None:
    fused = True
  if fused and lib.ldm_groupnorm_fused_supported(B, HW, C, groups, code(x.dtype)):
    assert out.dtype == x.dtype
    check(lib.ldm_groupnorm_fused(_ptr(x), row_ld(x), _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")),
                                  _ptr(out), row_ld(out), B, HW, C, groups, float(eps), int(bool(silu)),
                                  code(x.dtype), _stream()), "ldm_groupnorm_fused")
    return out
  nch = lib.ldm_groupnorm_nchunks(B, HW, C)
  if partial is None:
    partial = torch.empty(B * nch * groups * 2, dtype=torch.float32, device=x.device)
  assert partial.numel() >= B * nch * groups * 2
  dt = code(x.dtype)
  assert out.dtype == x.dtype
  check(lib.ldm_groupnorm_partial(_ptr(x), row_ld(x), _ptr(partial), B, HW, C, groups, nch, dt,
                                  _stream()), "ldm_groupnorm_partial")
  check(lib.ldm_groupnorm_apply(_ptr(x), row_ld(x), _ptr(partial), _ptr(_f32(gamma, "gamma")),
                                _ptr(_f32(beta, "beta")), _ptr(out), row_ld(out), B, HW, C, groups,
                                nch, float(eps), int(bool(silu)), dt, _stream()),
        "ldm_groupnorm_apply")
  return out


def layernorm(x, gamma, beta, out, eps=1e-5):
  Cc = x.shape[-1]
  rows = x.numel() // Cc
  assert out.dtype == x.dtype
  check(lib.ldm_layernorm(_ptr(x), row_ld(x), _ptr(_f32(gamma, "gamma")), _ptr(_f32(beta, "beta")),
                          _ptr(out), row_ld(out), rows, Cc, float(eps), code(x.dtype), _stream()),
        "ldm_layernorm")
  return out


def softmax_rows(x, out, scale=1.0):
  cols = x.shape[-1]
  rows = x.numel() // cols
  check(lib.ldm_softmax_rows(_ptr(x), row_ld(x), code(x.dtype), _ptr(out), row_ld(out),
                             code(out.dtype), rows, cols, float(scale), _stream()),
        "ldm_softmax_rows")
  return out


def attention(q, k, vt, out, heads, sp, scale, matrix_softmax=False):
  """q [R, Tq, heads*sp], k [R, Tk, heads*sp], vt [R, heads*sp, ldvt>=Tk], out like q.
  `matrix_softmax`: ldm_attention_ms (bf16, 40-wide heads padded to 48): q pre-scaled into the exp2
  domain, k[..., 40] = 1 and V^T row 40 = 1 per head (layout.MS_DIM; the projections' biases put them
  there), `scale` is not used."""
  R, Tq = q.shape[0], q.shape[1]
  Tk = k.shape[1]
  assert q.dtype == k.dtype == vt.dtype == out.dtype
  assert vt.shape[1] == heads * sp and vt.stride(2) == 1
  if matrix_softmax:
    check(lib.ldm_attention_ms(_ptr(q), q.stride(1), q.stride(0), _ptr(k), k.stride(1), k.stride(0),
                               _ptr(vt), vt.stride(1), vt.stride(0), _ptr(out), out.stride(1),
                               out.stride(0), R, heads, Tq, Tk, sp, code(q.dtype), _stream()), "ldm_attention_ms")
    return out
  check(lib.ldm_attention(_ptr(q), q.stride(1), q.stride(0), _ptr(k), k.stride(1), k.stride(0),
                          _ptr(vt), vt.stride(1), vt.stride(0), _ptr(out), out.stride(1),
                          out.stride(0), R, heads, Tq, Tk, sp, float(scale), code(q.dtype),
                          _stream()), "ldm_attention")
  return out


def ffn_geglu_supported(x):
  Cc = x.shape[-1]
  return bool(lib.ldm_ffn_geglu_supported(x.numel() // Cc, Cc, code(x.dtype)))


def ffn_geglu(x, w1, aux, w2, b2, out, eps):
  """out = x + b2 + W2 (a * gelu(g)), (a | g) = W1 LayerNorm(x) + b1: the transformer block's feed-forward as
  one launch per 128-row panel (ldm_ffn_geglu: bf16, C = 320).  w1 / aux from layout.ln_fold + layout.ffn_aux."""
  Cc = x.shape[-1]
  M = x.numel() // Cc
  assert out.dtype == x.dtype and tuple(out.shape) == tuple(x.shape)
  assert tuple(w1.shape) == (8 * Cc, Cc) and tuple(w2.shape) == (Cc, 4 * Cc) and w1.is_contiguous() and w2.is_contiguous()
  assert aux.dtype == torch.float32 and aux.numel() == 8 * Cc * 2 and aux.is_contiguous()
  check(lib.ldm_ffn_geglu(_ptr(x), row_ld(x), _ptr(w1), _ptr(aux), _ptr(w2), _ptr(_f32(b2, "b2")), _ptr(out),
                          row_ld(out), M, Cc, float(eps), code(x.dtype), _stream()), "ldm_ffn_geglu")
  return out


def st_tail(att, wo, bo, r0, w1, aux, w2, b2, wp, bp, r1, out, eps):
  """out = r1 + bp + Wp y,  y = feed-forward(h),  h = r0 + bo + Wo att: the o-projection of the cross-attention, the
  transformer block's feed-forward and the SpatialTransformer's proj_out as ONE row-panel launch (ldm_st_tail:
  bf16, C = 320, attention width 384).  h and y never reach HBM."""
  Cc = out.shape[-1]
  K0 = att.shape[-1]
  M = out.numel() // Cc
  assert att.numel() // K0 == M and att.dtype == out.dtype == r0.dtype == r1.dtype
  assert tuple(wo.shape) == (Cc, K0) and tuple(wp.shape) == (Cc, Cc) and wo.is_contiguous() and wp.is_contiguous()
  assert tuple(w1.shape) == (8 * Cc, Cc) and tuple(w2.shape) == (Cc, 4 * Cc) and w1.is_contiguous() and w2.is_contiguous()
  assert aux.dtype == torch.float32 and aux.numel() == 8 * Cc * 2 and aux.is_contiguous()
  check(lib.ldm_st_tail(_ptr(att), row_ld(att), K0, _ptr(wo), _ptr(_f32(bo, "bo")), _ptr(r0), row_ld(r0), _ptr(w1),
                        _ptr(aux), _ptr(w2), _ptr(_f32(b2, "b2")), _ptr(wp), _ptr(_f32(bp, "bp")), _ptr(r1), row_ld(r1),
                        _ptr(out), row_ld(out), M, Cc, float(eps), code(out.dtype), _stream()), "ldm_st_tail")
  return out


def st_xtail(q, ctx_k, ctx_vt, wo, bo, r0, w1, aux, w2, b2, wp, bp, r1, out, eps):
  """st_tail with the cross-attention in front: q [R, T, 384] are the queries, ctx_k [R, Tk, 384] / ctx_vt
  [R, 384, ld] the context keys / values^T, all in the matrix-side-softmax layout of ops.attention(...,
  matrix_softmax=True); the attention output lives only in LDS (ldm_st_xtail)."""
  Cc, K0 = out.shape[-1], q.shape[-1]
  R, T = q.shape[0], q.shape[1]
  M = R * T
  assert q.dim() == 3 and q.is_contiguous() and ctx_k.is_contiguous() and ctx_vt.is_contiguous()
  assert ctx_k.shape[0] == R and ctx_k.shape[2] == K0 and tuple(ctx_vt.shape[:2]) == (R, K0)
  assert out.numel() // Cc == M and q.dtype == out.dtype == r0.dtype == r1.dtype == ctx_k.dtype == ctx_vt.dtype
  assert tuple(wo.shape) == (Cc, K0) and tuple(wp.shape) == (Cc, Cc) and wo.is_contiguous() and wp.is_contiguous()
  assert tuple(w1.shape) == (8 * Cc, Cc) and tuple(w2.shape) == (Cc, 4 * Cc) and w1.is_contiguous() and w2.is_contiguous()
  assert aux.dtype == torch.float32 and aux.numel() == 8 * Cc * 2 and aux.is_contiguous()
  check(lib.ldm_st_xtail(_ptr(q), K0, K0, _ptr(ctx_k), _ptr(ctx_vt), ctx_k.shape[1], ctx_vt.shape[2], T, _ptr(wo),
                         _ptr(_f32(bo, "bo")), _ptr(r0), row_ld(r0), _ptr(w1), _ptr(aux), _ptr(w2), _ptr(_f32(b2, "b2")),
                         _ptr(wp), _ptr(_f32(bp, "bp")), _ptr(r1), row_ld(r1), _ptr(out), row_ld(out), M, Cc, float(eps),
                         code(out.dtype), _stream()), "ldm_st_xtail")
  return out


def st_block(att, wo1, bo1, r0, wq, qcs, qb, ctx_k, ctx_vt, wo2, bo2, w1, aux, w2, b2, wp, bp, r1, out, eps):
  """From the self-attention's output to the SpatialTransformer's output in ONE launch (ldm_st_block): o-projection
  + residual r0, LayerNorm-folded query projection, cross-attention against ctx_k / ctx_vt, o-projection + residual,
  feed-forward, proj_out + residual r1.  att [R, T, 384]; layouts as st_xtail / linear(ln_fold=...).
  `out` / the context may cover TWICE the rows of att / r0 / r1 (a classifier-free-guidance pair whose two halves
  are still identical in front of this launch): output row m >= R T reads input row m - R T."""
  Cc, K0 = out.shape[-1], att.shape[-1]
  Rin, T = att.shape[0], att.shape[1]
  R = ctx_k.shape[0]
  M = R * T
  assert R in (Rin, 2 * Rin) and r0.numel() // Cc == Rin * T and r1.numel() // Cc == Rin * T
  assert att.dim() == 3 and att.is_contiguous() and ctx_k.is_contiguous() and ctx_vt.is_contiguous()
  assert ctx_k.shape[0] == R and ctx_k.shape[2] == K0 and tuple(ctx_vt.shape[:2]) == (R, K0)
  assert out.numel() // Cc == M and att.dtype == out.dtype == r0.dtype == r1.dtype == ctx_k.dtype == ctx_vt.dtype
  for w_, shp in ((wo1, (Cc, K0)), (wq, (K0, Cc)), (wo2, (Cc, K0)), (wp, (Cc, Cc)), (w1, (8 * Cc, Cc)), (w2, (Cc, 4 * Cc))):
    assert tuple(w_.shape) == shp and w_.is_contiguous() and w_.dtype == out.dtype
  assert aux.dtype == torch.float32 and aux.numel() == 8 * Cc * 2 and aux.is_contiguous()
  assert qcs.dtype == torch.float32 and qcs.numel() == K0
  check(lib.ldm_st_block(_ptr(att), K0, K0, _ptr(wo1), _ptr(_f32(bo1, "bo1")), _ptr(r0), row_ld(r0), _ptr(wq), _ptr(qcs),
                         _ptr(_f32(qb, "qb")), _ptr(ctx_k), _ptr(ctx_vt), ctx_k.shape[1], ctx_vt.shape[2], T, _ptr(wo2),
                         _ptr(_f32(bo2, "bo2")), _ptr(w1), _ptr(aux), _ptr(w2), _ptr(_f32(b2, "b2")), _ptr(wp),
                         _ptr(_f32(bp, "bp")), _ptr(r1), row_ld(r1), _ptr(out), row_ld(out), M, Rin * T, Cc, float(eps),
                         code(out.dtype), _stream()), "ldm_st_block")
  return out


def time_embedding(out, channels, t_rows=None, steps=None, index=None):
  rows = out.shape[0]
  check(lib.ldm_time_embedding(_ptr(t_rows), _ptr(steps), _ptr(index), _ptr(_f32(out, "out")), rows,
                               channels, _stream()), "ldm_time_embedding")
  return out


def select_row(table, index, out, pre_decrement=False):
  """out[0, :] = table[*index, :] (float32); `pre_decrement`: *index is decremented first (the DDIM loop's
  device-side counter moves in the step's first launch)."""
  assert table.dim() == 2 and table.stride(1) == 1 and out.is_contiguous() and out.numel() == table.shape[1]
  assert index.dtype == torch.int32
  check(lib.ldm_select_row(_ptr(_f32(table, "table")), table.stride(0), table.shape[0], table.shape[1], _ptr(index),
                           int(bool(pre_decrement)), _ptr(_f32(out, "out")), _stream()), "ldm_select_row")
  return out


def gemv(x, wt, bias, y, act_in=ACT_NONE, act_out=ACT_NONE):
  """y[r, n] = act_out(sum_k act_in(x[r, k]) wt[n, k] + bias[n]); x, y float32."""
  rows, K = x.shape
  N = wt.shape[0]
  assert wt.shape[1] == K and wt.is_contiguous()
  check(lib.ldm_gemv(_ptr(_f32(x, "x")), x.stride(0), _ptr(wt), _ptr(_f32(bias, "bias")),
                     _ptr(_f32(y, "y")), y.stride(0), rows, N, K, act_in, act_out, code(wt.dtype),
                     _stream()), "ldm_gemv")
  return y


def cfg_ddim_update(eps_all, xt, xt_out, coef, index, guidance_scale, noise=None, x_unet_out=None,
                    dec_index=False, clip_denoised=False, noise_index_stride=0, pred_x0_out=None):
  B = xt.shape[0]
  n = xt.numel() // B
  xd = code(x_unet_out.dtype) if x_unet_out is not None else F32
  check(lib.ldm_cfg_ddim_update(_ptr(_f32(eps_all, "eps_all")), _ptr(_f32(xt, "xt")),
                                _ptr(_f32(noise, "noise")), int(noise_index_stride),
                                _ptr(_f32(xt_out, "xt_out")), _ptr(_f32(pred_x0_out, "pred_x0_out")),
                                _ptr(x_unet_out), xd, _ptr(_f32(coef, "coef")), _ptr(index),
                                int(bool(dec_index)), float(guidance_scale),
                                int(bool(clip_denoised)), B, n, _stream()), "ldm_cfg_ddim_update")
  return xt_out


def post_quant(latents, scale_factor, kernel_io, bias, out):
  Cc = latents.shape[-1]
  check(lib.ldm_post_quant(_ptr(_f32(latents, "latents")), float(scale_factor),
                           _ptr(_f32(kernel_io, "kernel")), _ptr(_f32(bias, "bias")), _ptr(out),
                           code(out.dtype), latents.numel() // Cc, Cc, _stream()), "ldm_post_quant")
  return out


def gaussian_sample(moments, out, noise=None, out_scale=1.0):
  Cc = out.shape[-1]
  assert moments.shape[-1] == 2 * Cc and moments.is_contiguous() and out.is_contiguous()
  assert noise is None or (noise.is_contiguous() and noise.numel() == out.numel())
  check(lib.ldm_gaussian_sample(_ptr(_f32(moments, "moments")), _ptr(_f32(noise, "noise")),
                                _ptr(_f32(out, "out")), float(out_scale), out.numel() // Cc, Cc,
                                _stream()), "ldm_gaussian_sample")
  return out


def vq_nearest(z, codebook, out, indices=None):
  Cc = z.shape[-1]
  check(lib.ldm_vq_nearest(_ptr(_f32(z, "z")), _ptr(_f32(codebook, "codebook")),
                           _ptr(_f32(out, "out")), _ptr(indices), z.numel() // Cc,
                           codebook.shape[0], Cc, _stream()), "ldm_vq_nearest")
  return out


def embedding(ids, tok_emb, pos_emb, out):
  rows, T = ids.shape
  assert ids.dtype == torch.int64 and ids.is_contiguous()
  check(lib.ldm_embedding(_ptr(ids), _ptr(_f32(tok_emb, "tok_emb")), _ptr(_f32(pos_emb, "pos_emb")),
                          _ptr(out), rows, T, tok_emb.shape[1], tok_emb.shape[0], code(out.dtype),
                          _stream()), "ldm_embedding")
  return out


def minmax_u8(x, out, scratch=None):
  B = x.shape[0]
  n = x.numel() // B
  assert x.is_contiguous() and out.dtype == torch.uint8
  if scratch is None:
    scratch = torch.empty(B * 128, dtype=torch.float32, device=x.device)
  check(lib.ldm_minmax_u8(_ptr(x), code(x.dtype), _ptr(out), _ptr(scratch), B, n, _stream()),
        "ldm_minmax_u8")
  return out


def cast(x, out):
  cols = x.shape[-1]
  rows = x.numel() // cols
  check(lib.ldm_cast(_ptr(x), row_ld(x), code(x.dtype), _ptr(out), row_ld(out), code(out.dtype),
                     rows, cols, _stream()), "ldm_cast")
  return out
