#!/usr/bin/env python3
"""conv3x3 -> GroupNorm + SiLU as the ResBlock launches them (unet.py:383-392), timed as a PAIR for every (tile,
split-K) plan of the convolution: a split plan leaves its reduce to the GroupNorm launch (ldm_groupnorm_splitk), so
only the pair's time ranks the plans.  16 rotating buffer sets, 16 pairs per captured graph.

    python tools/conv_gn_pair_probe.py --rows 16 --hw 32 --cin 320 --cout 320
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_tf2_amd import ops  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--rows", type=int, default=16)
  ap.add_argument("--hw", type=int, default=32)
  ap.add_argument("--cin", type=int, default=320)
  ap.add_argument("--cout", type=int, default=320)
  ap.add_argument("--sets", type=int, default=8)
  args = ap.parse_args()
  dev, bf = torch.device("cuda:0"), torch.bfloat16
  R, H, Ci, Co = args.rows, args.hw, args.cin, args.cout
  sets = []
  for _ in range(args.sets):
    x = torch.randn(R, H, H, Ci, device=dev).to(bf)
    w = (torch.randn(Co, 9 * Ci, device=dev) * (9 * Ci) ** -0.5).to(bf)
    sets.append((x, w, torch.empty(R, H, H, Co, device=dev, dtype=bf), torch.empty(R, H, H, Co, device=dev, dtype=bf)))
  bias, temb = torch.randn(Co, device=dev), torch.randn(R, Co, device=dev)
  gamma, beta = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
  ws = ops.new_workspace(dev)

  def pair(s, tile, split):
    x, w, y, g = sets[s]
    with ops.workspace_scope(ws):
      pend = ops.conv3x3(x, w, y, bias=bias, addend=temb, tile=tile, split_k=split, defer_reduce=True)
      ops.groupnorm(y, gamma, beta, g, 1e-5, silu=True, pending=pend, store_x=False)

  def timed(tile, split):
    fns = [(lambda s=s: pair(s, tile, split)) for s in range(args.sets)] * 2
    for f in fns:
      f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
      for f in fns:
        f()
    gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      gr.replay()
      e1.record()
      e1.synchronize()
      best = min(best, e0.elapsed_time(e1) / len(fns))
    return best * 1e3

  M = R * H * H
  gf = 2.0 * M * Co * 9 * Ci / 1e9
  print(f"# conv {H}x{H} {Ci}->{Co}, {R} rows: M={M}, {gf:.1f} GFLOP; conv + GroupNorm pair, us")
  for tile in (9, 10, 15, 11, 12, 2, 13, 0):
    for split in ((1, 2, 3, 4) if tile not in (13, 15) else (1, 2) if tile == 15 else (1,)):
      try:
        t = timed(tile, split)
      except RuntimeError as e:
        print(f"  tile {tile:2d} split {split}: -- {str(e)[:70]}")
        continue
      print(f"  tile {tile:2d} split {split}: {t:7.1f} us")


if __name__ == "__main__":
  main()
