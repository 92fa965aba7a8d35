#!/usr/bin/env python3
"""Calibrates the float32 parity gate ("Fixed tolerance"): runs the CPU
oracle in float64 and in float32 on the same seeded weights/inputs and prints the drift
between them -- the noise floor that any correct float32 implementation (the reference's TF
CPU kernels, this oracle, the HIP path) sits on.  The GPU gate in tests/ is a small multiple
of these numbers (see DESIGN.md "Tolerance").

    python tools/calibrate_tolerance.py [--full]     # --full adds one full-size U-Net eval
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_tf2_amd import weights as Wt  # noqa: E402
from oracle import ldm_oracle as O  # noqa: E402


def rel(a, b):
  a, b = a.double(), b.double()
  return ((a - b).norm() / b.norm()).item(), (a - b).abs().max().item()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--full", action="store_true")
  args = ap.parse_args()
  torch.set_num_threads(min(8, os.cpu_count() or 1))
  ucfg = dict(model_channels=64, out_channels=4, num_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8)
  tcfg = dict(vocab_size=1000, encoder_stack_size=2, hidden_size=128, num_heads=4, size_per_head=32,
              max_seq_len=77, filter_size=256)
  kcfg = dict(latent_channels=4, channels=64, num_blocks=2, multipliers=(1, 2, 4, 4))
  w = {"unet": Wt.init_weights(Wt.unet_manifest(context_dim=128, **ucfg), seed=2, mode="random", scope="unet"),
       "cond_stage_model": Wt.init_weights(Wt.transformer_manifest(**tcfg), seed=2, mode="random",
                                           scope="cond_stage_model"),
       "autoencoder": Wt.init_weights(Wt.decoder_manifest(**kcfg), seed=2, mode="random", scope="autoencoder")}
  g = torch.Generator().manual_seed(0)
  x = torch.randn(4, 16, 16, 4, generator=g)
  ctx = torch.randn(4, 77, 128, generator=g)
  y32 = O.unet_forward(x, [981] * 4, ctx, w["unet"], dtype=torch.float32)
  y64 = O.unet_forward(x, [981] * 4, ctx, w["unet"], dtype=torch.float64)
  print("tiny U-Net, one evaluation      f32 vs f64: rel %.3e  maxabs %.3e" % rel(y32, y64))
  z = torch.randn(2, 8, 8, 4, generator=g)
  d32 = O.decoder_forward(z, w["autoencoder"], dtype=torch.float32)
  d64 = O.decoder_forward(z, w["autoencoder"], dtype=torch.float64)
  print("tiny KL decoder                 f32 vs f64: rel %.3e  maxabs %.3e" % rel(d32, d64))
  ids = torch.randint(0, 1000, (4, 77), generator=g)
  xT = torch.randn(2, 16, 16, 4, generator=g)
  for n in (10, 50):
    ldm = dict(num_steps=1000, beta_start=0.00085, beta_end=0.012, v_posterior=0., scale_factor=0.18215,
               eta=0., num_ddim_steps=n)
    i32 = O.ddim_p_sample_loop(ids, xT, w, ldm, guidance_scale=5., dtype=torch.float32)
    i64 = O.ddim_p_sample_loop(ids, xT, w, ldm, guidance_scale=5., dtype=torch.float64)
    print("tiny free-running loop, N=%-3d    f32 vs f64: rel %.3e  maxabs %.3e" % ((n,) + rel(i32, i64)))
  if args.full:
    t0 = time.time()
    wu = Wt.init_weights(Wt.unet_manifest(), seed=2, scope="unet")
    x = torch.randn(2, 32, 32, 4, generator=g)
    ctx = torch.randn(2, 77, 1280, generator=g)
    y32 = O.unet_forward(x, [981, 981], ctx, wu, dtype=torch.float32)
    y64 = O.unet_forward(x, [981, 981], ctx, wu, dtype=torch.float64)
    print("full-size U-Net (872 M), 1 eval f32 vs f64: rel %.3e  maxabs %.3e   (%.0f s)"
          % (rel(y32, y64) + (time.time() - t0,)))


if __name__ == "__main__":
  main()
