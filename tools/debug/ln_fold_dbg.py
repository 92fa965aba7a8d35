import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ldm_tf2_amd import ops as o, layout as L
from oracle import ldm_oracle as O
dev = torch.device("cuda:0")
BF = torch.bfloat16
M, K, N = 512, 320, 768
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, K, generator=g) * (0.5 + 2 * torch.rand(M, 1, generator=g))).to(BF)
w = torch.randn(N, K, generator=g) * K ** -0.5
for name, gamma, beta, bias in [("id", torch.ones(K), torch.zeros(K), torch.zeros(N)),
                                ("gamma", 1 + 0.3 * torch.randn(K, generator=g), torch.zeros(K), torch.zeros(N)),
                                ("beta", torch.ones(K), 0.2 * torch.randn(K, generator=g), torch.zeros(N)),
                                ("bias", torch.ones(K), torch.zeros(K), torch.randn(N, generator=g))]:
  ref = O.dense(O.layer_norm(x.float(), gamma, beta, eps=1e-5), w.t(), bias)
  wq, cs, bb = L.ln_fold(w, gamma.numpy(), beta.numpy(), bias.numpy(), BF, dev)
  out = torch.full((M, N), float("nan"), dtype=BF, device=dev)
  o.linear(x.to(dev), wq, out, bias=bb, ln_fold=(cs, 1e-5))
  got = out.float().cpu()
  err = (got - ref)
  print(name, "rel", (err.norm() / ref.norm()).item())
  # per-row scale: least-squares got ~ a_m * ref
  a = (got * ref).sum(1) / (ref * ref).sum(1)
  print("  row scale a: min %.4f max %.4f; rows 0..7" % (a.min(), a.max()), a[:8].numpy().round(4), a[64:72].numpy().round(4))
  res = got - a[:, None] * ref
  print("  residual after row scale rel", (res.norm() / ref.norm()).item())
  rowerr = err.norm(dim=1) / ref.norm(dim=1)
  print("  rowerr by 16-row block:", rowerr.reshape(-1, 16).mean(1)[:16].numpy().round(3))
  colerr = err.norm(dim=0) / ref.norm(dim=0)
  print("  colerr by 16-col block:", colerr.reshape(-1, 16).mean(1)[:16].numpy().round(3))
