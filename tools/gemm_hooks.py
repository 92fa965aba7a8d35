"""Measurement hooks around the MFMA GEMM family, kept OUT of the product module.

Every ldm_gemm launch of the package goes through `ldm_tf2_amd.ops._gemm` (/ `_gemm_deferred`), the
row-panel launches that chain several products (ldm_ffn_geglu, ldm_st_tail, ldm_st_xtail, ldm_st_block)
through the ops functions of the same names; the context managers below replace those functions for the
duration of a measurement and restore them afterwards.  Used by bench.py (roofline leg) and
tools/step_breakdown.py only.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import torch

from ldm_tf2_amd import ops
from ldm_tf2_amd._lib import lib


# row-panel op -> (position of `out` in its arguments, K of the [M, 320] x [320, K] product with the same FLOPs at
# C = 320: feed-forward 2560 + 1280, o-projection 384, proj_out 320, self-attention o-projection 384, query 384)
PANEL_OPS = {"ffn_geglu": (5, 3840), "st_tail": (11, 4544), "st_xtail": (13, 4544), "st_block": (18, 5312)}


@contextlib.contextmanager
def skip_gemms(counter):
  """ldm_gemm launches are NOT enqueued, only counted in counter[0].  bench.py captures one
  U-Net step this way and times it against the full step: the difference is the MFMA
  GEMM/conv family's share of a step (HIP events on graph replays, no per-launch overhead).
  Outputs of such a step are garbage by construction."""
  orig, orig_d = ops._gemm, ops._gemm_deferred

  def _skip(p, device):
    counter[0] += 1

  def _skip_deferred(p, device, out, keep):
    # no slabs are left behind: the GroupNorm that would have completed the product runs in its plain
    # form, so what the fused launch spends on the slabs counts as the family's time
    counter[0] += 1
    return None

  def _skip_panel(*a, **k):
    # the row-panel launches are products of the same family (their FLOPs are in the family's count); the one
    # with the cross-attention inside carries that attention's time along, which only lowers the quoted rate
    counter[0] += 1

  panels = {n: getattr(ops, n) for n in PANEL_OPS}
  ops._gemm, ops._gemm_deferred = _skip, _skip_deferred
  for n in PANEL_OPS:
    setattr(ops, n, _skip_panel)
  try:
    yield counter
  finally:
    ops._gemm, ops._gemm_deferred = orig, orig_d
    for n, f in panels.items():
      setattr(ops, n, f)


@contextlib.contextmanager
def time_gemms(sink):
  """Every ldm_gemm launch is bracketed by HIP events on the launch stream; appends
  (start, end, problem key, (M, N, K, batch, act, dtype, tile, split_k)) to `sink`."""
  orig, orig_d = ops._gemm, ops._gemm_deferred

  def _bracket(p, device, call):
    key = ops.plan_key(p)
    ops.resolve_plan(p)
    ws = ops.workspace(device)
    p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel()
    t, s = C.c_int(), C.c_int()
    lib.ldm_gemm_plan(C.byref(p), C.byref(t), C.byref(s))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = call()
    e1.record()
    sink.append((e0, e1, key, (p.M, p.N, p.K, p.batch, p.act, p.dtype, t.value, s.value)))
    return r

  def _timed(p, device):
    return _bracket(p, device, lambda: orig(p, device))

  def _timed_deferred(p, device, out, keep):
    # (the bracket holds the main kernel only: the reduce runs inside the consuming GroupNorm launch)
    return _bracket(p, device, lambda: orig_d(p, device, out, keep))

  def _panel(name, f):
    def timed(*a, **k):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      r = f(*a, **k)
      e1.record()
      out = a[PANEL_OPS[name][0]]
      M = out.numel() // out.shape[-1]
      sink.append((e0, e1, f"{name} M={M}", (M, out.shape[-1], PANEL_OPS[name][1], 1, 0, ops.code(out.dtype), 0, 1)))
      return r
    return timed

  panels = {n: getattr(ops, n) for n in PANEL_OPS}
  ops._gemm, ops._gemm_deferred = _timed, _timed_deferred
  for n, f in panels.items():
    setattr(ops, n, _panel(n, f))
  try:
    yield sink
  finally:
    ops._gemm, ops._gemm_deferred = orig, orig_d
    for n, f in panels.items():
      setattr(ops, n, f)
