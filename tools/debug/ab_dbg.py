import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN
from ldm_tf2_amd import weights as Wt
from ldm_tf2_amd.unet import UNet
dev = torch.device("cuda:0")
cfg = BN.FULL["unet"]
w = Wt.init_weights(Wt.unet_manifest(**cfg), seed=2, scope="unet")
for B in (8, 16):
  R = 2 * B
  g = np.random.default_rng(0)
  x = torch.from_numpy(g.standard_normal((R, 32, 32, 4)).astype(np.float32)).to(dev)
  ctx = torch.from_numpy(g.standard_normal((R, 77, 1280)).astype(np.float32)).to(dev)
  t = torch.full((R,), 500, dtype=torch.int32, device=dev)
  ref = None
  for name, kw in [("base", dict(fold_layernorm=False, defer_reduce=False)), ("fold", dict(defer_reduce=False)),
                   ("defer", dict(fold_layernorm=False)), ("both", {}), ("both_again", {})]:
    unet = UNet(**cfg, weights=w, dtype=torch.bfloat16, device=dev, **kw)
    unet.set_context(ctx)
    outs = []
    for rep in range(3):
      out = torch.empty(R, 32, 32, 4, device=dev)
      unet.forward(x, t_rows=t, out=out, shared_t=True)
      torch.cuda.synchronize()
      outs.append(out.clone())
    o = outs[0]
    if ref is None:
      ref = o.double()
    print(f"B={B} {name:>10s}: norm {o.double().norm().item():.4f} nan {int(torch.isnan(o).sum())} "
          f"rel vs base {((o.double() - ref).norm() / ref.norm()).item():.3e}  rep1==rep0 {torch.equal(outs[0], outs[1])} rep2==rep0 {torch.equal(outs[0], outs[2])}", flush=True)
    del unet
