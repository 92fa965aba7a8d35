#!/usr/bin/env python3
"""Timing of the persistent kernel (tile 13/14) on one problem; LDM_G3_DEBUG ablations are read once
per process, so run one process per variant."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.toolslib  # noqa: F401  (ablations live in the tools build only)
from ldm_tf2_amd import ops
from tools.gemm_bench import time_fn
M, K, N, tile = (int(v) for v in sys.argv[1:5])
geglu = len(sys.argv) > 5 and sys.argv[5] == "geglu"
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
b = torch.randn(N, device=dev)
out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.bfloat16)
ms = time_fn(lambda: ops.linear(x, w, out, bias=b, tile=tile, act=ops.ACT_GEGLU if geglu else ops.ACT_NONE), 3)
tag = " geglu" if geglu else ""
print(f"dbg={os.environ.get('LDM_G3_DEBUG', '0')} M={M} K={K} N={N} tile={tile}{tag}: {ms * 1e3:.1f} us {2.0 * M * N * K / 1e9 / ms:.0f} TF/s")
