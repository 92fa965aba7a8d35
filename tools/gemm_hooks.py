"""Measurement hooks around ldm_gemm, kept OUT of the product module.

Every ldm_gemm launch of the package goes through `ldm_tf2_amd.ops._gemm`; the context
managers below replace that one function for the duration of a measurement and restore it
afterwards.  Used by bench.py (roofline leg) and tools/step_breakdown.py only.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import torch

from ldm_tf2_amd import ops
from ldm_tf2_amd._lib import lib


@contextlib.contextmanager
def skip_gemms(counter):
  """ldm_gemm launches are NOT enqueued, only counted in counter[0].  bench.py captures one
  U-Net step this way and times it against the full step: the difference is the MFMA
  GEMM/conv family's share of a step (HIP events on graph replays, no per-launch overhead).
  Outputs of such a step are garbage by construction."""
  orig = ops._gemm

  def _skip(p, device):
    counter[0] += 1

  ops._gemm = _skip
  try:
    yield counter
  finally:
    ops._gemm = orig


@contextlib.contextmanager
def time_gemms(sink):
  """Every ldm_gemm launch is bracketed by HIP events on the launch stream; appends
  (start, end, problem key, (M, N, K, batch, act, dtype, tile, split_k)) to `sink`."""
  orig = ops._gemm

  def _timed(p, device):
    key = ops.plan_key(p)
    ops.resolve_plan(p)
    ws = ops.workspace(device)
    p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel()
    t, s = C.c_int(), C.c_int()
    lib.ldm_gemm_plan(C.byref(p), C.byref(t), C.byref(s))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(p, device)
    e1.record()
    sink.append((e0, e1, key, (p.M, p.N, p.K, p.batch, p.act, p.dtype, t.value, s.value)))

  ops._gemm = _timed
  try:
    yield sink
  finally:
    ops._gemm = orig
