#!/usr/bin/env python3
"""One conv shape on one tile, 20 launches: the workload for a `rocprofv3 --pmc` pass that asks where the
waves of the ping-pong conv kernels spend their cycles (tools/profile: SQ_WAVE_CYCLES, SQ_WAIT_ANY,
SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, SQ_VALU_MFMA_BUSY_CYCLES, ...).

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -- python3 tools/conv_pmc_probe.py 16 1280 1280 15
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_tf2_amd import ops  # noqa: E402

hw, cin, cout, tile = (int(v) for v in sys.argv[1:5])
R = 32
dev = torch.device("cuda:0")
x = torch.randn(R, hw, hw, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(R, hw, hw, cout, device=dev, dtype=torch.bfloat16)
for _ in range(20):
  ops.conv3x3(x, w, out, tile=tile)
torch.cuda.synchronize()
print("done")
