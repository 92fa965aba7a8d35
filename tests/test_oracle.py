"""CPU tests: the oracle against the reference's own pins (tests/golden) and its
L1 op restatements against torch.nn.functional."""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

from ldm_tf2_amd import weights as Wt
from oracle import ldm_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_step_tables_and_schedule_kats():
  g = json.load(open(os.path.join(GOLD, "schedule_kats.json")))
  for n in ("10", "50", "200"):
    s = O.make_schedule(1000, 0.00085, 0.012, 0.0, int(n))
    assert s["ddim_steps"].dtype == np.int32
    assert s["ddim_steps"].tolist() == g["ddim_steps"][n]          # bit-exact
  s = O.make_schedule(1000, 0.00085, 0.012, 0.0, 50)
  assert s["ddim_steps"][-1] == 981            # convert_ckpt_pytorch_to_tf2.py:402
  assert abs(s["alphas_cumprod"][0] - g["alphas_cumprod_0"]) < 1e-12
  assert abs(s["alphas_cumprod"][981] - g["alphas_cumprod_981"]) < 1e-12
  # a_prev at index 0 is abar[0], not 1 (model_runners.py:412-415)
  assert np.float32(s["ddim_alphas_cumprod_prev"][0]) == np.float32(s["alphas_cumprod"][0])
  assert np.all(s["ddim_sigmas"] == 0)


def test_param_totals():
  # README.md:33 (~0.87B / ~0.54B / ~0.09B)
  assert Wt.count_params(Wt.unet_manifest()) == 872300484
  assert Wt.count_params(Wt.transformer_manifest()) == 542895360
  assert Wt.count_params(Wt.decoder_manifest()) == 49490199


def test_time_embedding_kat():
  e = O.get_time_embedding([981], 320)
  assert torch.allclose(e[0, :3], torch.tensor([0.67996, -0.79843, 0.57811]), atol=1e-4)
  assert torch.allclose(e[0, 160:163], torch.tensor([0.73325, 0.60209, 0.81596]), atol=1e-4)


def test_ops_vs_torch_functional():
  g = torch.Generator().manual_seed(0)
  x = torch.randn(2, 8, 8, 64, generator=g)
  gamma, beta = torch.randn(64, generator=g), torch.randn(64, generator=g)
  ref = F.group_norm(x.permute(0, 3, 1, 2), 32, gamma, beta, eps=1e-5).permute(0, 2, 3, 1)
  assert torch.allclose(O.group_norm(x, gamma, beta, eps=1e-5), ref, atol=1e-5)
  assert torch.allclose(O.layer_norm(x, gamma, beta), F.layer_norm(x, (64,), gamma, beta, 1e-5), atol=1e-5)
  assert torch.allclose(O.gelu(x), F.gelu(x), atol=1e-6)
  assert torch.allclose(O.silu(x), F.silu(x), atol=1e-6)
  k = torch.randn(3, 3, 64, 32, generator=g)
  ref = F.conv2d(x.permute(0, 3, 1, 2), k.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
  assert torch.allclose(O.conv2d(x, k, None), ref, atol=1e-4)
  up = O.upsample_nearest2x(x)
  assert torch.equal(up, F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1))
  assert torch.equal(up[:, 5, 7], x[:, 2, 3])


def test_tiny_unet_f32_vs_f64():
  m = Wt.unet_manifest(model_channels=64, context_dim=128)
  w = Wt.init_weights(m, mode="random")
  g = np.random.default_rng(0)
  x = g.standard_normal((2, 8, 8, 4)).astype(np.float32)
  ctx = g.standard_normal((2, 77, 128)).astype(np.float32)
  y32 = O.unet_forward(x, [981, 981], ctx, w)
  y64 = O.unet_forward(x, [981, 981], ctx, w, dtype=torch.float64)
  assert y32.shape == (2, 8, 8, 4)
  assert (y32 - y64).abs().max() < 1e-4


def test_tensor_to_image():
  a = np.array([[[[0.0, 1.0, 0.5]]], [[[-1.0, 3.0, 1.0]]]], dtype=np.float32)
  out = O.tensor_to_image(a)
  assert out.dtype == np.uint8
  assert out[0].ravel().tolist() == [0, 255, 127]
  assert out[1].ravel().tolist() == [0, 255, 127]
