import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ldm_tf2_amd import ops as o, layout as L
dev = torch.device("cuda:0")
BF = torch.bfloat16
M, K, N = 512, 320, 256
x = torch.zeros(M, K)
for m in range(M):
  x[m, m % K] = 1.0
g = torch.Generator().manual_seed(0)
w = torch.randn(N, K, generator=g)
w = w - w.mean(1, keepdim=True)
wq, cs, bb = L.ln_fold(w, np.ones(K, np.float32), np.zeros(K, np.float32), np.zeros(N, np.float32), BF, dev)
out = torch.zeros(M, N, dtype=BF, device=dev)
o.linear(x.to(BF).to(dev), wq, out, bias=bb, ln_fold=(cs, 1e-5))
got = out.float().cpu()
wf = wq.float().cpu()
# ratio = rstd the kernel used for row m
rs = torch.tensor([(got[m] * wf[:, m % K]).sum() / (wf[:, m % K] ** 2).sum() for m in range(M)])
print("expected 17.86 (element counted once), 316 (never), 12.6 (twice)")
np.set_printoptions(linewidth=200)
print(rs.reshape(-1, 16).numpy().round(1))
