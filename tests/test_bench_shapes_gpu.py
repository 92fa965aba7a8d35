"""Parity at the BENCHMARKED shapes and launch plans ("configs untested").

(1) Every key of every packaged plan table (ldm_tf2_amd/plans/*.json): the exact problem the
    key names (M, N, K, conv geometry, stride, upsample, activation, dtype) on random data,
    run with the table's forced (tile, split-K) and with the cost model's choice, compared with
    the oracle (conv2d / dense, GEGLU) at the tolerances of tests/test_ops_gpu.py.  The oracle
    is evaluated on a bounded set of rows / images (first, middle, last); the remaining rows of
    the planned launch are checked against the auto launch (both HIP, same data), so every
    output row of every planned launch is covered.
(2) Full-size U-Net, ONE evaluation at the bench batch shapes: R=32 bf16 (C3), R=16 bf16 (C4),
    R=8 f32 at latent 64x64 (C5) with DISTINCT rows; the oracle evaluates 2-3 of the rows
    (rows are independent: unet.py has no cross-sample op), the HIP result must match on those
    rows, with the packaged plan table of that configuration active (UNet.forward selects it).
(3) Full-size KL decode at B=4, latent 64x64 (C5): one image row against the oracle.
"""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ldm_tf2_amd import ops, weights as Wt  # noqa: E402
from ldm_tf2_amd._lib import ACT_GEGLU, BF16  # noqa: E402
from oracle import ldm_oracle as O  # noqa: E402

TOL = {torch.float32: dict(rtol=2e-4, atol=2e-4), torch.bfloat16: dict(rtol=3e-2, atol=3e-2)}
REL = {torch.float32: 5e-5, torch.bfloat16: 4e-2}


def _plan_cases():
  d = os.path.join(os.path.dirname(ops.__file__), "plans")
  seen, out = set(), []
  for path in sorted(glob.glob(os.path.join(d, "*.json"))):
    for key, plan in json.load(open(path))["plans"].items():
      if (key, tuple(plan)) in seen:
        continue
      seen.add((key, tuple(plan)))
      out.append(pytest.param(key, tuple(plan), id=f"{key.replace(' ', '_')}-t{plan[0]}s{plan[1]}"))
  return out


def _fields(key):
  return dict((k.rstrip("0123456789"), int(k[len(k.rstrip("0123456789")):])) for k in key.split())


def _rand(shape, dtype, seed, scale=1.0):
  g = torch.Generator().manual_seed(seed)
  return (torch.randn(*shape, generator=g) * scale).to(dtype)


def _close(got, ref, dtype, what):
  got, ref = got.detach().float().cpu(), ref.float()
  tol = TOL[dtype]
  err = (got - ref).abs().max().item()
  assert torch.allclose(got, ref, rtol=tol["rtol"], atol=tol["atol"]), f"{what}: max err {err}"


@pytest.mark.parametrize("key,plan", _plan_cases())
def test_every_packaged_plan_matches_the_oracle(dev, key, plan):
  f = _fields(key)
  dtype = torch.bfloat16 if f["dt"] == BF16 else torch.float32
  tile, split = plan
  M, N, K = f["M"], f["N"], f["K"]
  torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
  if f["conv"]:
    H, W, s, up = f["H"], f["W"], f["s"], bool(f["u"])
    # " x2": the ResBlock's second convolution (Cin = Cout = N) with its 1x1 shortcut over Cin2 = K - 9 N more
    # channels of a second image as extra K columns (ldm_gemm a2)
    Cin, Cin2 = (N, K - 9 * N) if f.get("x") else (K // 9, 0)
    OH = (2 * H if up else H) // s
    B = M // (OH * OH)
    assert B * OH * OH == M and not f["nlp"]
    x = _rand((B, H, W, Cin), dtype, 1)
    k = _rand((3, 3, Cin, N), dtype, 2, (9 * Cin) ** -0.5)
    bias = _rand((N,), torch.float32, 3)
    wt = k.permute(3, 0, 1, 2).reshape(N, 9 * Cin).contiguous()
    x2 = ks = x2d = None
    if Cin2:
      x2 = _rand((B, H, W, Cin2), dtype, 4)
      ks = _rand((Cin2, N), dtype, 5, Cin2 ** -0.5)
      wt = torch.cat([wt, ks.t()], 1).contiguous()
      x2d = x2.to(dev)
    wt = wt.to(dev)
    xd, bd = x.to(dev), bias.to(dev)
    outs = {}
    for name, (t, sp) in (("plan", (tile, split)), ("auto", (0, 0))):
      out = torch.zeros(B, OH, OH, N, dtype=dtype, device=dev)
      ops.conv3x3(xd, wt, out, bias=bd, stride=s, upsample=up, tile=t, split_k=sp, x2=x2d)
      outs[name] = out
    rows = sorted({0, B // 2, B - 1})
    xs = x[rows].float()
    ref = O.conv2d(O.upsample_nearest2x(xs) if up else xs, k.float(), bias, stride=s)
    if Cin2:
      ref = ref + O.dense(x2[rows].float(), ks.float(), None)
    for name, out in outs.items():
      _close(out[rows], ref, dtype, f"{name} conv {key}")
    _close(outs["plan"], outs["auto"].cpu(), dtype, f"plan vs auto {key}")
  else:
    x = _rand((M, K), dtype, 1)
    w = _rand((N, K), dtype, 2, K ** -0.5)
    bias = _rand((N,), torch.float32, 3)
    geglu = f["act"] == ACT_GEGLU
    nout = N // 2 if geglu else N
    xd, wd, bd = x.to(dev), w.to(dev), bias.to(dev)
    outs = {}
    for name, (t, sp) in (("plan", (tile, split)), ("auto", (0, 0))):
      out = torch.zeros(M, nout, dtype=dtype, device=dev)
      ops.linear(xd, wd, out, bias=bd, act=f["act"], tile=t, split_k=sp)
      outs[name] = out
    rows = torch.cat([torch.arange(0, min(M, 256)), torch.arange(M // 2, min(M, M // 2 + 128)),
                      torch.arange(max(0, M - 256), M)]).unique()
    y = x[rows].float() @ w.float().t() + bias
    if geglu:
      yv = y.reshape(len(rows), N // 64, 2, 32)      # device layout: 32 value rows then their 32 gate rows
      y = (yv[:, :, 0] * O.gelu(yv[:, :, 1])).reshape(len(rows), N // 2)
    for name, out in outs.items():
      _close(out[rows.to(dev)], y, dtype, f"{name} gemm {key}")
    _close(outs["plan"], outs["auto"].cpu(), dtype, f"plan vs auto {key}")


# ---- full-size U-Net at the bench batch shapes ---------------------------------------------------
UNET = dict(model_channels=320, out_channels=4, num_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8)
KL = dict(latent_channels=4, channels=128, num_blocks=2, multipliers=(1, 2, 4, 4))
TXT = dict(vocab_size=30522, encoder_stack_size=32, hidden_size=1280, num_heads=8, size_per_head=64,
           filter_size=5120, max_seq_len=77)
LDM = dict(num_steps=1000, beta_start=0.00085, beta_end=0.012, scale_factor=0.18215, eta=0.)


def _rel(got, ref):
  got = got.detach().float().cpu().double()
  ref = ref.detach().double()
  return ((got - ref).norm() / ref.norm()).item()


@pytest.fixture(scope="module")
def unet_w():
  return Wt.init_weights(Wt.unet_manifest(**UNET), seed=2, mode="random", scope="unet")


@pytest.mark.parametrize("R,latent,dtype,check_rows", [
    (32, 32, torch.bfloat16, (0, 17, 31)),     # C3: B=16 -> 32 U-Net rows
    (16, 32, torch.bfloat16, (0, 9, 15)),      # C4: B=8 per GPU
    (8, 64, torch.float32, (0, 5)),            # C5: B=4 at latent 64x64 (f32)
    (8, 64, torch.bfloat16, (7,)),             # C5 in bf16 (the bench's --dtype bf16 --latent 64 variant)
    (8, 32, torch.float32, (3, 4)),            # C2: B=4 f32
])
def test_fullsize_unet_at_bench_batch(dev, unet_w, R, latent, dtype, check_rows):
  """Distinct rows (own x, own context, shared t as in the DDIM loop); the packaged plan table of
  (R, latent, dtype) is the one UNet.forward activates, so this runs the benchmarked launches."""
  from ldm_tf2_amd.unet import UNet
  g = np.random.default_rng(100 + R + latent)
  x = g.standard_normal((R, latent, latent, 4)).astype(np.float32)
  ctx = g.standard_normal((R, 77, 1280)).astype(np.float32)
  t = np.full((R,), 701, np.int32)
  unet = UNet(**UNET, weights=unet_w, dtype=dtype, device=dev)
  table = ops.gemm_plans(R, latent, dtype)
  got = unet(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(ctx))
  assert tuple(got.shape) == (R, latent, latent, 4) and bool(torch.isfinite(got).all())
  torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
  rows = list(check_rows)
  with torch.no_grad():
    ref = O.unet_forward(x[rows], t[rows], ctx[rows], unet_w)
  r = _rel(got[rows], ref)
  print(f"full-size U-Net R={R} latent={latent} [{dtype}] rows {rows}: rel={r:.3e} (plan table entries: {len(table)})")
  assert r < REL[dtype]
  # the shared-t fast path of the sampling loop (one temb row for all rows) gives the same result
  xd = torch.from_numpy(x).to(dev)
  got2 = unet.forward(xd, t_rows=torch.from_numpy(t).to(dev), shared_t=True)
  assert _rel(got2, got.cpu()) < (1e-6 if dtype == torch.float32 else 1e-2)


def test_fullsize_decode_b4_latent64(dev):
  """C5's decode: B=4 at latent 64x64 -> 512x512 (decoder convs at 64^2 .. 512^2)."""
  from ldm_tf2_amd.autoencoder import AutoencoderKL
  kl_w = Wt.init_weights(Wt.decoder_manifest(**KL), seed=2, mode="random", scope="autoencoder")
  z = np.random.default_rng(5).standard_normal((4, 64, 64, 4)).astype(np.float32)
  torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
  with torch.no_grad():
    ref = O.decoder_forward(torch.from_numpy(z[2:3]) / 0.18215, kl_w)
  for dtype in (torch.float32, torch.bfloat16):
    ae = AutoencoderKL(**KL, weights=kl_w, dtype=dtype, device=dev)
    got = ae.decode(torch.from_numpy(z), scale_factor=0.18215)
    assert tuple(got.shape) == (4, 512, 512, 3)
    r = _rel(got[2:3], ref)
    print(f"full-size decode B=4 latent 64 [{dtype}] image 2: rel={r:.3e}")
    assert r < REL[dtype]
    del ae
    torch.cuda.empty_cache()


def test_c3_loop_reproducible_bf16(dev, unet_w):
  """The benchmarked loop itself (BASELINE configs[2] shape: B=16, bf16, CFG 5, captured step graph,
  packaged plan table, persistent / ping-pong / split-K launches) at 6 DDIM steps: two passes are
  bit-identical, the eager (uncaptured) loop gives the same bits, and a sampler with ANOTHER prompt
  in between does not disturb it -- a race in a hand-scheduled kernel shows up here as a flipped bit."""
  from ldm_tf2_amd.autoencoder import AutoencoderKL
  from ldm_tf2_amd.model_runners import LatentDiffusionModelSampler
  from ldm_tf2_amd.transformer import TransformerModel
  from ldm_tf2_amd.unet import UNet
  dt = torch.bfloat16
  small_txt = dict(TXT, encoder_stack_size=2)
  wt = Wt.init_weights(Wt.transformer_manifest(**small_txt), seed=2, scope="cond_stage_model")
  kl_w = Wt.init_weights(Wt.decoder_manifest(**KL), seed=2, mode="random", scope="autoencoder")
  unet = UNet(**UNET, weights=unet_w, dtype=dt, device=dev)
  ae = AutoencoderKL(**KL, weights=kl_w, dtype=dt, device=dev)
  txt = TransformerModel(**small_txt, weights=wt, dtype=dt, device=dev)
  B = 16

  def ids(seed):
    cond = np.random.default_rng(seed).integers(0, 30522, size=(B, 77))
    return np.concatenate([np.tile([[101, 102] + [0] * 75], (B, 1)), cond], 0)

  s = LatentDiffusionModelSampler(unet, ae, txt, verbose=False, use_graph=True, num_ddim_steps=6, **LDM)
  a = s.ddim_p_sample_loop(ids(1), [B, 32, 32, 4], 5., seed=3).clone()
  assert tuple(a.shape) == (B, 256, 256, 3) and bool(torch.isfinite(a.float()).all())
  other = s.ddim_p_sample_loop(ids(2), [B, 32, 32, 4], 5., seed=3).clone()
  assert not torch.equal(a, other)
  b = s.ddim_p_sample_loop(ids(1), [B, 32, 32, 4], 5., seed=3).clone()
  assert torch.equal(a, b)
  e = LatentDiffusionModelSampler(unet, ae, txt, verbose=False, use_graph=False, num_ddim_steps=6, **LDM)
  c = e.ddim_p_sample_loop(ids(1), [B, 32, 32, 4], 5., seed=3)
  assert torch.equal(a, c)
