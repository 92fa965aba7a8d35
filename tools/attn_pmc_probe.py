#!/usr/bin/env python3
"""20 launches of the 32x32-level self-attention (R = 32, T = 1024, 8 heads of 40 -> 48), matrix-side softmax
form unless --plain: workload for rocprofv3 --pmc passes (tools/pmc_kernel_avg.py averages the counters)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_tf2_amd import ops  # noqa: E402

plain = "--plain" in sys.argv
R, T, H, d, sp = 32, 1024, 8, 40, 48
dev = torch.device("cuda:0")
q = torch.randn(R, T, H, sp, device=dev)
k = torch.randn(R, T, H, sp, device=dev)
v = torch.randn(R, H, sp, T, device=dev)
q[..., d:] = 0
k[..., d:] = 0
v[:, :, d:, :] = 0
if not plain:
  q = q * (d ** -0.5 * 1.4426950408889634)
  k[..., d] = 1
  v[:, :, d, :] = 1
q, k, v = (t.to(torch.bfloat16) for t in (q.reshape(R, T, H * sp), k.reshape(R, T, H * sp), v.reshape(R, H * sp, T)))
o = torch.empty_like(q)
for _ in range(20):
  ops.attention(q, k, v, o, H, sp, d ** -0.5, matrix_softmax=not plain)
torch.cuda.synchronize()
print("done")
