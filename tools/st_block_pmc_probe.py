#!/usr/bin/env python3
"""20 ldm_st_block launches at the benchmark size (R = 32 samples x 1024 tokens, C = 320): workload for a
rocprofv3 --pmc pass (tools/pmc_kernel_avg.py <counter_collection.csv> st_tail_kernel averages the counters)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_tf2_amd import layout as L, ops  # noqa: E402

dev, bf = torch.device("cuda:0"), torch.bfloat16
C, K0, R, T = 320, 384, 32, 1024
M = R * T
g = torch.Generator().manual_seed(0)
rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
att = rn(R, T, K0).to(bf).to(dev)
r0, r1 = rn(M, C).to(bf).to(dev), rn(M, C).to(bf).to(dev)
wo1, wo2 = (rn(C, K0, sc=K0 ** -0.5).to(bf).to(dev) for _ in range(2))
wp = rn(C, C, sc=C ** -0.5).to(bf).to(dev)
b = torch.zeros(C, device=dev)
one, zero = np.ones(C, np.float32), np.zeros(C, np.float32)
wq, qcs, qb = L.ln_fold(rn(K0, C, sc=C ** -0.5), one, zero, None, bf, dev)
gw, gb = L.geglu_kernel(rn(C, 8 * C, sc=C ** -0.5).numpy(), rn(8 * C).numpy(), torch.float32, "cpu")
w1, cs, bb = L.ln_fold(gw, one, zero, gb.numpy(), bf, dev)
aux = L.ffn_aux(cs, bb)
w2 = rn(C, 4 * C, sc=(4 * C) ** -0.5).to(bf).to(dev)
ck, cv = rn(R, 77, K0).to(bf).to(dev), rn(R, K0, 80).to(bf).to(dev)
for t_ in (ck.view(R, 77, 8, 48),):
  t_[..., 40:] = 0
  t_[..., 40] = 1
cv.view(R, 8, 48, 80)[:, :, 40:, :] = 0
cv.view(R, 8, 48, 80)[:, :, 40, :] = 1
out = torch.empty(M, C, dtype=bf, device=dev)
for _ in range(20):
  ops.st_block(att, wo1, b, r0, wq, qcs, qb, ck, cv, wo2, b, w1, aux, w2, b, wp, b, r1, out, 1e-5)
torch.cuda.synchronize()
print("done")
