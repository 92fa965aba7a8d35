#!/usr/bin/env python3
"""20 GEGLU launches on the persistent kernel (M, K, N from argv): workload for a rocprofv3 --pmc pass."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_tf2_amd import ops  # noqa: E402

M, K, N = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
b = torch.randn(N, device=dev)
out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
for _ in range(20):
  ops.linear(x, w, out, bias=b, tile=14, act=ops.ACT_GEGLU)
torch.cuda.synchronize()
print("done")
