#!/usr/bin/env python3
"""ldm_ffn_geglu (LayerNorm -> GEGLU -> FF-out + residual as one row-panel launch) against the two launches it
replaces, M rows of C = 320, graph-replayed best of 5 (tools.gemm_bench.time_fn).

    python tools/ffn_probe.py [M ...]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("LDM_FFN_DEBUG"):             # timing ablations: tools build only (make tools)
  import tools.toolslib  # noqa: F401,E402
from ldm_tf2_amd import layout as L, ops  # noqa: E402
from tools.gemm_bench import time_fn  # noqa: E402

dev = torch.device("cuda:0")
C = 320
g = torch.Generator().manual_seed(0)
k1 = (torch.randn(C, 8 * C, generator=g) * C ** -0.5).numpy()
b1 = torch.randn(8 * C, generator=g).numpy()
k2 = (torch.randn(4 * C, C, generator=g) * (4 * C) ** -0.5).numpy()
b2 = torch.randn(C, generator=g).to(dev)
gw, gb = L.geglu_kernel(k1, b1, torch.float32, "cpu")
w1, cs, bb = L.ln_fold(gw, np.ones(C, np.float32), np.zeros(C, np.float32), gb.numpy(), torch.bfloat16, dev)
w2 = L.dense_kernel(k2, torch.bfloat16, dev)
wo = L.dense_kernel((torch.randn(384, C, generator=g) * 384 ** -0.5).numpy(), torch.bfloat16, dev)
wp = L.dense_kernel((torch.randn(C, C, generator=g) * C ** -0.5).numpy(), torch.bfloat16, dev)
aux = L.ffn_aux(cs, bb)
for M in [int(a) for a in sys.argv[1:]] or [32768, 16384]:
  x = torch.randn(M, C, device=dev).to(torch.bfloat16)
  out = torch.empty_like(x)
  ff = torch.empty(M, 4 * C, dtype=torch.bfloat16, device=dev)
  t_f = time_fn(lambda: ops.ffn_geglu(x, w1, aux, w2, b2, out, 1e-5), 5)

  def two():
    ops.linear(x, w1, ff, bias=bb, act=ops.ACT_GEGLU, ln_fold=(cs, 1e-5), tile=14)
    ops.linear(ff, w2, out, bias=b2, residual=x, tile=13)

  t_2 = time_fn(two, 5)
  gf = 2.0 * M * C * 8 * C * 1e-9 + 2.0 * M * 4 * C * C * 1e-9
  print(f"dbg={os.environ.get('LDM_FFN_DEBUG', '0')} M={M}: fused {t_f * 1e3:7.1f} us ({gf / t_f:5.0f} TFLOP/s)   two launches {t_2 * 1e3:7.1f} us ({gf / t_2:5.0f} TFLOP/s)")
  att = torch.randn(M, 384, device=dev).to(torch.bfloat16)
  r1 = torch.randn(M, C, device=dev).to(torch.bfloat16)
  h, y = torch.empty_like(x), torch.empty_like(x)
  t_t = time_fn(lambda: ops.st_tail(att, wo, b2, x, w1, aux, w2, b2, wp, b2, r1, out, 1e-5), 5)

  def four():
    ops.linear(att, wo, h, bias=b2, residual=x)
    ops.linear(h, w1, ff, bias=bb, act=ops.ACT_GEGLU, ln_fold=(cs, 1e-5), tile=14)
    ops.linear(ff, w2, y, bias=b2, residual=h, tile=13)
    ops.linear(y, wp, out, bias=b2, residual=r1)

  t_4 = time_fn(four, 5)
  gf += 2.0 * M * C * (384 + C) * 1e-9
  print(f"      st_tail (o-projection + feed-forward + proj_out): {t_t * 1e3:7.1f} us ({gf / t_t:5.0f} TFLOP/s)   four launches {t_4 * 1e3:7.1f} us")
  if M % 1024 == 0:
    R = M // 1024
    qx = torch.randn(R, 1024, 384, device=dev).to(torch.bfloat16)
    ck = torch.randn(R, 77, 384, device=dev).to(torch.bfloat16)
    cv = torch.randn(R, 384, 80, device=dev).to(torch.bfloat16)
    for t_ in (qx.view(R, 1024, 8, 48), ck.view(R, 77, 8, 48)):
      t_[..., 40:] = 0
    ck.view(R, 77, 8, 48)[..., 40] = 1
    cv.view(R, 8, 48, 80)[:, :, 40:, :] = 0
    cv.view(R, 8, 48, 80)[:, :, 40, :] = 1
    a2 = torch.empty_like(qx)
    t_x = time_fn(lambda: ops.st_xtail(qx, ck, cv, wo, b2, x, w1, aux, w2, b2, wp, b2, r1, out, 1e-5), 5)

    def att_tail():
      ops.attention(qx, ck, cv, a2, 8, 48, 40 ** -0.5, matrix_softmax=True)
      ops.st_tail(a2, wo, b2, x, w1, aux, w2, b2, wp, b2, r1, out, 1e-5)

    t_x2 = time_fn(att_tail, 5)
    print(f"      st_xtail (cross-attention + tail): {t_x * 1e3:7.1f} us   attention + st_tail {t_x2 * 1e3:7.1f} us")
    wq, qcs, qb = L.ln_fold(torch.randn(384, C, generator=g) * C ** -0.5, np.ones(C, np.float32), np.zeros(C, np.float32), None, torch.bfloat16, dev)
    a1 = att.view(R, 1024, 384)
    t_b = time_fn(lambda: ops.st_block(a1, wo, b2, x, wq, qcs, qb, ck, cv, wo, b2, w1, aux, w2, b2, wp, b2, r1, out, 1e-5), 5)

    def gemms_xtail():
      ops.linear(att, wo, h, bias=b2, residual=x)
      ops.linear(h, wq, qx.view(M, 384), bias=qb, ln_fold=(qcs, 1e-5))
      ops.st_xtail(qx, ck, cv, wo, b2, h, w1, aux, w2, b2, wp, b2, r1, out, 1e-5)

    t_b2 = time_fn(gemms_xtail, 5)
    print(f"      st_block (o-projection + q projection + st_xtail): {t_b * 1e3:7.1f} us   2 GEMMs + st_xtail {t_b2 * 1e3:7.1f} us")
