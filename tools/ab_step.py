#!/usr/bin/env python3
"""A/B of U-Net step variants in ONE process on ONE device: each variant's step (U-Net forward on 2B
rows, full size, random-init weights) is captured as a HIP graph, and the graphs are replayed in
interleaved rounds (variant order rotated) -- timings of different pool boxes differ by several %,
so only same-process numbers rank two launch forms.

    python tools/ab_step.py --batch 16 [--latent 32] [--rounds 6] name=kw:val,kw:val ...
e.g. python tools/ab_step.py base=fold_layernorm:0,defer_reduce:0 fold=defer_reduce:0 both=
(values: ints; a variant's kwargs go to the UNet constructor; `plans:<file>` captures the variant with that launch-plan
table instead of the packaged ones)
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BN  # noqa: E402
from ldm_tf2_amd import ops, weights as Wt  # noqa: E402
from ldm_tf2_amd.unet import UNet  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--batch", type=int, default=16)
  ap.add_argument("--latent", type=int, default=32)
  ap.add_argument("--rounds", type=int, default=6)
  ap.add_argument("--reps", type=int, default=10)
  ap.add_argument("variants", nargs="+")
  args = ap.parse_args()
  dev = torch.device("cuda:0")
  cfg = BN.FULL["unet"]
  w = Wt.init_weights(Wt.unet_manifest(**cfg), seed=2, scope="unet")
  R = 2 * args.batch
  g = np.random.default_rng(0)
  xh = g.standard_normal((args.batch, args.latent, args.latent, 4)).astype(np.float32)
  x = torch.from_numpy(np.concatenate([xh, xh], 0)).to(dev)       # the DDIM loop's concat([xt, xt]) (model_runners.py:449)
  ctx = torch.from_numpy(g.standard_normal((R, 77, 1280)).astype(np.float32)).to(dev)
  t = torch.full((R,), 500, dtype=torch.int32, device=dev)
  graphs, names, outs = [], [], []
  for spec in args.variants:
    name, _, kws = spec.partition("=")
    kw = {}
    plans = None
    temb = 1
    for item in filter(None, kws.split(",")):
      k, _, v = item.partition(":")
      if k == "plans":              # this variant's launch-plan table (a tuner output) instead of the packaged ones
        plans = v
      elif k == "temb":             # the step's temb projections as one row-select from a per-loop table (1) or as four launches (0)
        temb = int(v)
      else:
        kw[k] = int(v)
    ops.clear_plans()               # (a captured graph keeps the plans it was captured with)
    if plans:
      ops.load_plans(plans)
    else:
      ops._load_default_plans()
    unet = UNet(**cfg, weights=w, dtype=torch.bfloat16, device=dev, **kw)
    unet.set_context(ctx)
    out = torch.empty(R, args.latent, args.latent, 4, device=dev)
    steps = torch.tensor([500], dtype=torch.int32, device=dev)
    idx = torch.zeros(1, dtype=torch.int32, device=dev)
    tkw = dict(steps=steps, index=idx, temb_table=unet.temb_table(steps).clone()) if temb else dict(t_rows=t, shared_t=True)
    unet.forward(x, out=out, paired_rows=True, **tkw)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
      unet.forward(x, out=out, paired_rows=True, **tkw)
    for _ in range(3):
      gr.replay()
    torch.cuda.synchronize()
    graphs.append((gr, unet, out))       # keep `out` alive: a freed output's address would be handed to the
    names.append(name)                   # next clone, and a later replay of this graph would overwrite that
    outs.append(out.clone())
    print(f"[ab] {name}: captured ({kw})", flush=True)
  ms = {n: [] for n in names}
  for r in range(args.rounds):
    order = list(range(len(names)))
    order = order[r % len(order):] + order[:r % len(order)]
    for i in order:
      gr = graphs[i][0]
      gr.replay()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(args.reps):
        gr.replay()
      e1.record()
      e1.synchronize()
      ms[names[i]].append(e0.elapsed_time(e1) / args.reps)
  ref = outs[0].double()
  for i, n in enumerate(names):
    a = np.array(ms[n])
    d = ((outs[i].double() - ref).norm() / ref.norm()).item()
    print(f"{n:>16s}: median {np.median(a):7.3f} ms  min {a.min():7.3f}  max {a.max():7.3f}   (output vs first variant: rel {d:.2e})")


if __name__ == "__main__":
  main()
