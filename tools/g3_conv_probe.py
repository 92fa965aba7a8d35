#!/usr/bin/env python3
"""Timing of the persistent kernel (tile 13) on one conv; LDM_G3_DEBUG ablations per process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.toolslib  # noqa: F401  (ablations live in the tools build only)
from ldm_tf2_amd import ops
from tools.gemm_bench import time_fn
R, hw, cin, cout, tile = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
x = torch.randn(R, hw, hw, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).to(torch.bfloat16)
b = torch.randn(cout, device=dev)
out = torch.empty(R, hw, hw, cout, device=dev, dtype=torch.bfloat16)
ms = time_fn(lambda: ops.conv3x3(x, w, out, bias=b, tile=tile), 3)
gf = 2.0 * R * hw * hw * cout * 9 * cin / 1e9
print(f"dbg={os.environ.get('LDM_G3_DEBUG', '0')} conv R={R} {hw}x{hw} {cin}->{cout} tile={tile}: {ms * 1e3:.1f} us {gf / ms:.0f} TF/s")
