"""First-stage ENCODE path: HIP encoder (through the C ABI)
against the CPU oracle -- the autoencoder's pad (0,1),(0,1) stride-2 downsample conv, the
KL posterior (mean | logvar -> sample / mode) and the VQ encoder + nearest-codebook lookup.

Tolerances as in test_models_gpu.py: relative L2 5e-5 (float32), 4e-2 (bfloat16);
VQ code indices: exact except where the two nearest codes are closer than the f32 noise.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ldm_tf2_amd import ops  # noqa: E402
from ldm_tf2_amd import layout as L  # noqa: E402
from ldm_tf2_amd import weights as Wt  # noqa: E402
from oracle import ldm_oracle as O  # noqa: E402

REL = {torch.float32: 5e-5, torch.bfloat16: 4e-2}
DT = [torch.float32, torch.bfloat16]
KL_CFG = dict(latent_channels=4, channels=64, num_blocks=2, multipliers=(1, 2, 4, 4))
VQ_CFG = dict(latent_channels=4, channels=64, num_blocks=2, multipliers=(1, 2, 2, 4),
              attention_resolutions=(8,), vocab_size=512)


def rel_err(got, ref):
  got = got.detach().float().cpu().double()
  ref = ref.detach().double()
  return ((got - ref).norm() / ref.norm()).item(), (got - ref).abs().max().item()


def check(got, ref, dtype, what, factor=1.0):
  r, m = rel_err(got, ref)
  print(f"{what} [{dtype}]: rel={r:.3e} maxabs={m:.3e}")
  assert r < REL[dtype] * factor, f"{what}: rel err {r:.3e} (max abs {m:.3e})"


def _kl_weights(cfg, image_size):
  m = Wt.decoder_manifest(**cfg)
  m.update(Wt.encoder_manifest(**cfg, image_size=image_size, double_z=True))
  return Wt.init_weights(m, seed=2, mode="random", scope="autoencoder")


@pytest.mark.parametrize("dtype", DT, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (1, 32, 48, 128, 64), (3, 8, 8, 64, 192)])
def test_downsample_conv_no_lead_pad(dev, dtype, shape):
  """autoencoder.py:133-136: pad [[0,1],[0,1]] then 3x3 stride-2 VALID."""
  B, H, W, Cin, Cout = shape
  g = torch.Generator().manual_seed(3)
  x = torch.randn(B, H, W, Cin, generator=g)
  k = torch.randn(3, 3, Cin, Cout, generator=g) * (9 * Cin) ** -0.5
  b = torch.randn(Cout, generator=g)
  xd = x.to(dev, dtype)
  xr = xd.float().cpu()
  kd = L.conv_kernel(k.numpy(), dtype, dev)
  kr = k.to(dtype).float()
  ref = O.conv2d(xr, kr, b, stride=2, pad=((0, 1), (0, 1)))
  out = torch.empty(B, H // 2, W // 2, Cout, dtype=dtype, device=dev)
  ops.conv3x3(xd, kd, out, bias=b.to(dev), stride=2, no_lead_pad=True)
  torch.cuda.synchronize()
  check(out, ref, dtype, f"downsample conv {shape}", factor=0.25 if dtype == torch.bfloat16 else 1.0)
  # and it is NOT the U-Net's pad (1,1),(1,1) variant
  out2 = torch.empty_like(out)
  ops.conv3x3(xd, kd, out2, bias=b.to(dev), stride=2)
  torch.cuda.synchronize()
  assert rel_err(out2, ref)[0] > 0.1


def test_gaussian_sample_kernel(dev):
  g = torch.Generator().manual_seed(5)
  mom = torch.randn(2, 4, 4, 8, generator=g) * 3
  mom[0, 0, 0, 4:] = torch.tensor([-50.0, 30.0, 0.0, 1.0])    # beyond the clip range on purpose
  noise = torch.randn(2, 4, 4, 4, generator=g)
  mean, logvar, sample = O.diagonal_gaussian(mom, noise)
  from ldm_tf2_amd.autoencoder import DiagonalGaussian
  post = DiagonalGaussian(mom.to(dev))
  assert torch.equal(post._logvar.cpu(), logvar) and torch.equal(post._mean.cpu(), mean)
  assert torch.equal(post.mode().cpu(), mean)
  got = post.sample(noise=noise).cpu()
  assert torch.allclose(got, sample, rtol=2e-6, atol=1e-6)
  # std from the UNCLIPPED logvar (distribution.py:18)
  assert abs(got[0, 0, 0, 1].item() - (mom[0, 0, 0, 1] + torch.exp(torch.tensor(15.0)) * noise[0, 0, 0, 1]).item()) \
      < 1e-5 * abs(got[0, 0, 0, 1].item())


@pytest.mark.parametrize("dtype", DT, ids=["f32", "bf16"])
def test_encode_kl(dev, dtype):
  from ldm_tf2_amd.autoencoder import AutoencoderKL
  w = _kl_weights(KL_CFG, 64)
  g = torch.Generator().manual_seed(11)
  img = torch.rand(2, 64, 64, 3, generator=g) * 2 - 1
  noise = torch.randn(2, 8, 8, 4, generator=g)
  ae = AutoencoderKL(**KL_CFG, weights=w, dtype=dtype, device=dev)
  post = ae.encode(img)
  torch.cuda.synchronize()
  ref = O.encoder_forward(img, w)
  check(post._moments, ref, dtype, "KL encoder moments")
  mean, logvar, sample = O.diagonal_gaussian(ref, noise)
  check(post.mode(), mean, dtype, "KL posterior mode")
  check(post.sample(noise=noise), sample, dtype, "KL posterior sample", factor=2.0)
  # autoencoder.py:343-351 call(): decode(sample) -- the reconstruction round trip runs
  rec = ae.decode(post.mode())
  torch.cuda.synchronize()
  check(rec, O.decoder_forward(mean, w), dtype, "KL reconstruction", factor=2.0)
  assert tuple(rec.shape) == (2, 64, 64, 3)


def test_encode_needs_encoder(dev):
  from ldm_tf2_amd.autoencoder import AutoencoderKL
  w = Wt.init_weights(Wt.decoder_manifest(**KL_CFG), seed=2, mode="random", scope="autoencoder")
  ae = AutoencoderKL(**KL_CFG, weights=w, dtype=torch.float32, device=dev)
  with pytest.raises(RuntimeError):
    ae.encode(torch.zeros(1, 64, 64, 3))


@pytest.mark.parametrize("dtype", DT, ids=["f32", "bf16"])
def test_encode_vq(dev, dtype):
  from ldm_tf2_amd.autoencoder import AutoencoderVQ
  m = Wt.decoder_manifest(**VQ_CFG, latent_size=8)
  m.update(Wt.encoder_manifest(**VQ_CFG, image_size=64, double_z=False))
  w = Wt.init_weights(m, seed=2, mode="random", scope="autoencoder")
  assert any(k.startswith("encoder/down/") and "/attention/" in k for k in m)   # size-8 DownBlocks attend
  g = torch.Generator().manual_seed(12)
  img = torch.rand(2, 64, 64, 3, generator=g) * 2 - 1
  ae = AutoencoderVQ(**VQ_CFG, latent_size=8, weights=w, dtype=dtype, device=dev)
  z = ae.encode(img, only_encode=True)
  q, loss, idx = ae.encode(img)
  torch.cuda.synchronize()
  zr, qr, lossr, idxr = O.vq_encode(img, w, attention_resolutions=(8,), beta=0.25)
  check(z, zr, dtype, "VQ encoder latents")
  if dtype == torch.float32:
    same = (idx.cpu() == idxr)
    print(f"VQ indices equal: {same.float().mean().item():.4f}")
    assert same.float().mean().item() >= 0.98
    assert torch.allclose(q.cpu().reshape(-1, 4)[same], qr.reshape(-1, 4)[same], atol=1e-5)
    assert abs(loss.item() - lossr.item()) < 1e-3 * abs(lossr.item())
  # self-consistency in either dtype: q is the codebook row idx names, for the z the GPU produced
  cb = torch.from_numpy(w["quantize/kernel"])
  zq, idx2 = O.vq_nearest(z.cpu(), cb)
  assert (idx.cpu() == idx2).float().mean().item() >= 0.98
  assert tuple(idx.shape) == (2 * 8 * 8,) and idx.dtype == torch.int64


def test_encode_kl_full_size(dev):
  """txt2img-f8 KL encoder (34.2 M parameters) on one 256x256 image, float32."""
  from ldm_tf2_amd.autoencoder import AutoencoderKL
  cfg = dict(latent_channels=4, channels=128, num_blocks=2, multipliers=(1, 2, 4, 4))
  m = Wt.decoder_manifest(**cfg)
  m.update(Wt.encoder_manifest(**cfg, image_size=256))
  assert Wt.count_params(m) == 83653863                       # the KL-f8 autoencoder's size
  w = Wt.init_weights(m, seed=2, scope="autoencoder")
  g = torch.Generator().manual_seed(13)
  img = torch.rand(1, 256, 256, 3, generator=g) * 2 - 1
  ae = AutoencoderKL(**cfg, weights=w, dtype=torch.float32, device=dev)
  post = ae.encode(img)
  torch.cuda.synchronize()
  ref = O.encoder_forward(img, w)
  assert tuple(ref.shape) == (1, 32, 32, 8)
  check(post._moments, ref, torch.float32, "full-size KL encoder moments")
