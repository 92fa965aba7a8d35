#!/usr/bin/env python3
"""Experiment: does running the U-Net step of two half-batches concurrently (two HIP graphs on
two streams) beat one full batch?  The step is a chain of ~450 dependent launches, about half
of them latency-bound; two independent chains can fill each other's gaps.

    python tools/dual_stream_bench.py [--batch 16] [--ddim-steps 50] [--ways 2]
Prints ms per U-Net step for: one sampler at B, one at B/ways, `ways` samplers at B/ways
replayed concurrently on their own streams.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BN  # noqa: E402
from ldm_tf2_amd import weights as Wt  # noqa: E402
from ldm_tf2_amd.autoencoder import AutoencoderKL  # noqa: E402
from ldm_tf2_amd.model_runners import LatentDiffusionModelSampler  # noqa: E402
from ldm_tf2_amd.transformer import TransformerModel  # noqa: E402
from ldm_tf2_amd.unet import UNet  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--batch", type=int, default=16)
  ap.add_argument("--ddim-steps", type=int, default=50)
  ap.add_argument("--ways", type=int, default=2)
  args = ap.parse_args()
  dev, dt, cfg = torch.device("cuda:0"), torch.bfloat16, BN.FULL
  w = {"unet": Wt.init_weights(Wt.unet_manifest(**cfg["unet"]), seed=2, scope="unet"),
       "cond_stage_model": Wt.init_weights(Wt.transformer_manifest(**cfg["cond_stage_model"]), seed=2,
                                           scope="cond_stage_model"),
       "autoencoder": Wt.init_weights(Wt.decoder_manifest(**cfg["autoencoder_kl"]), seed=2, scope="autoencoder")}
  txt = TransformerModel(**cfg["cond_stage_model"], weights=w["cond_stage_model"], dtype=dt, device=dev)
  ae = AutoencoderKL(**cfg["autoencoder_kl"], weights=w["autoencoder"], dtype=dt, device=dev)
  ldm = dict(cfg["ldm"], num_ddim_steps=args.ddim_steps)
  n = args.ddim_steps

  def sampler(B):
    unet = UNet(**cfg["unet"], weights=w["unet"], dtype=dt, device=dev)
    s = LatentDiffusionModelSampler(unet, ae, txt, use_graph=True, verbose=False, **ldm)
    s.ddim_p_sample_loop(BN.synthetic_token_ids(B), [B, 32, 32, 4], guidance_scale=5., seed=0)
    torch.cuda.synchronize()
    return s

  def run(samplers, streams):
    for s in samplers:
      s._index_dev.fill_(n - 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
      for s, st in zip(samplers, streams):
        with torch.cuda.stream(st):
          s._graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

  full = sampler(args.batch)
  cur = torch.cuda.current_stream()
  print(f"B={args.batch:3d} x1 : {min(run([full], [cur]) for _ in range(3)):7.3f} ms/step")
  del full
  part = [sampler(args.batch // args.ways) for _ in range(args.ways)]
  print(f"B={args.batch // args.ways:3d} x1 : {min(run(part[:1], [cur]) for _ in range(3)):7.3f} ms/step")
  streams = [torch.cuda.Stream() for _ in part]
  print(f"B={args.batch // args.ways:3d} x{args.ways} concurrent: {min(run(part, streams) for _ in range(3)):7.3f} ms/step")


if __name__ == "__main__":
  main()
