"""Round-2 regression / coverage tests on the GPU (all through the C ABI, against the oracle).

* a second, different prompt on ONE sampler must not reuse the first prompt's cross-attention
  K/V: sample A then B, check B against the oracle and a fresh sampler;
* SURVEY 8f N1 on the GPU: a CompVis-named state dict -> checkpoint.from_compvis_state_dict ->
  HIP U-Net + decoder + text encoder, against the oracle fed the same converted weights;
* AutoencoderKL.call / AutoencoderVQ.call (autoencoder.py:344-351, :438-444);
* the real VQ-f8 decoder configuration (16384 codes, multipliers (1,2,2,4), attention at 32);
* decoder batch chunking: the progressive sampler at FULL size decodes B * N/record_freq frames
  (beyond ldm_gemm's 2 GiB operand limit in one call) -- must run and match the per-frame decode;
* two samplers replaying on two streams do not share split-K workspaces.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ldm_tf2_amd import weights as Wt  # noqa: E402
from oracle import ldm_oracle as O  # noqa: E402

REL = {torch.float32: 5e-5, torch.bfloat16: 4e-2}
UNET_CFG = dict(model_channels=64, out_channels=4, num_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8)
CTX_DIM = 128
TXT_CFG = dict(vocab_size=1000, encoder_stack_size=2, hidden_size=CTX_DIM, num_heads=4,
               size_per_head=32, max_seq_len=77, filter_size=256)
KL_CFG = dict(latent_channels=4, channels=64, num_blocks=2, multipliers=(1, 2, 4, 4))
LDM = dict(num_steps=1000, beta_start=0.00085, beta_end=0.012, v_posterior=0., scale_factor=0.18215,
           eta=0., num_ddim_steps=10)


def rel(got, ref):
  got = got.detach().float().cpu().double()
  ref = ref.detach().double() if isinstance(ref, torch.Tensor) else torch.as_tensor(ref).double()
  return ((got - ref).norm() / ref.norm()).item()


def _weights():
  return dict(
      unet=Wt.init_weights(Wt.unet_manifest(context_dim=CTX_DIM, **UNET_CFG), seed=2, mode="random", scope="unet"),
      cond_stage_model=Wt.init_weights(Wt.transformer_manifest(**TXT_CFG), seed=2, mode="random", scope="cond_stage_model"),
      autoencoder=Wt.init_weights(Wt.decoder_manifest(**KL_CFG), seed=2, mode="random", scope="autoencoder"))


def _sampler(dev, dtype, w, ldm=LDM, **kw):
  from ldm_tf2_amd.autoencoder import AutoencoderKL
  from ldm_tf2_amd.model_runners import LatentDiffusionModelSampler
  from ldm_tf2_amd.transformer import TransformerModel
  from ldm_tf2_amd.unet import UNet
  return LatentDiffusionModelSampler(
      UNet(**UNET_CFG, weights=w["unet"], dtype=dtype, device=dev, context_dim=CTX_DIM),
      AutoencoderKL(**KL_CFG, weights=w["autoencoder"], dtype=dtype, device=dev),
      TransformerModel(**TXT_CFG, weights=w["cond_stage_model"], dtype=dtype, device=dev), verbose=False, **ldm, **kw)


def _ids(B, seed):
  cond = np.random.default_rng(seed).integers(0, 1000, size=(1, 77))
  return np.concatenate([np.tile([[101, 102] + [0] * 75], (B, 1)), np.tile(cond, (B, 1))], 0)


@pytest.mark.parametrize("use_graph", [True, False])
def test_second_prompt_is_not_served_from_the_first_prompts_context(dev, use_graph):
  """Prompt A then prompt B on one sampler (same shapes, so the caching allocator hands B's
  context the address A's had): B must equal the oracle's B and a fresh sampler's B."""
  w = _weights()
  B = 2
  x_T = np.random.default_rng(3).standard_normal((B, 16, 16, 4)).astype(np.float32)
  s = _sampler(dev, torch.float32, w, use_graph=use_graph)
  ids_a, ids_b = _ids(B, 1), _ids(B, 2)
  img_a = s.ddim_p_sample_loop(ids_a, [B, 16, 16, 4], 5., x_T=x_T).clone()
  img_b = s.ddim_p_sample_loop(ids_b, [B, 16, 16, 4], 5., x_T=x_T).clone()
  ref_b = O.ddim_p_sample_loop(ids_b, x_T, w, LDM, guidance_scale=5., num_heads=8)
  assert rel(img_b, ref_b) < 5 * REL[torch.float32], "prompt B was sampled with prompt A's cross-attention K/V"
  fresh = _sampler(dev, torch.float32, w, use_graph=use_graph).ddim_p_sample_loop(ids_b, [B, 16, 16, 4], 5., x_T=x_T)
  assert torch.equal(img_b, fresh)
  assert rel(img_a, ref_b) > 1e-2                      # the two prompts really differ
  # the U-Net operator alone, called twice with different contexts (unet.py:118 contract)
  g = np.random.default_rng(5)
  x = g.standard_normal((4, 16, 16, 4)).astype(np.float32)
  t = np.array([981, 981, 21, 500], np.int32)
  unet = s._unet
  for seed in (1, 2):
    ctx = np.random.default_rng(seed).standard_normal((4, 77, CTX_DIM)).astype(np.float32)
    got = unet(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(ctx))
    assert rel(got, O.unet_forward(x, t, ctx, w["unet"])) < REL[torch.float32]
  # ddim_sample after the operator call: the sampler re-projects ITS context
  ctx = np.random.default_rng(9).standard_normal((2 * B, 77, CTX_DIM)).astype(np.float32)
  sched = O.make_schedule(1000, 0.00085, 0.012, 0., 10)
  ref, _, _ = O.ddim_sample(x_T, torch.from_numpy(ctx), 7, sched, w["unet"], guidance_scale=5., clip_denoised=True)
  got = s.ddim_sample(torch.from_numpy(x_T), torch.from_numpy(ctx), 7, guidance_scale=5., clip_denoised=True)
  assert rel(got, ref) < 4 * REL[torch.float32]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_compvis_named_weights_through_the_loader(dev, dtype):
  """N1: CompVis state-dict names / layouts -> from_compvis_state_dict -> the three HIP models,
  against the oracle on the same converted weights (the converter's key list, order and
  transposes themselves are pinned by tests/test_converter_pin_cpu.py)."""
  from ldm_tf2_amd import checkpoint as C
  w = _weights()
  ucfg = dict(UNET_CFG, context_dim=CTX_DIM)
  acfg = dict(KL_CFG)
  sd = C.to_compvis_state_dict(w, ucfg, TXT_CFG, acfg)
  # a PyTorch checkpoint stores conv OIHW / Linear [out, in]; scramble memory order too
  sd = {k: np.asfortranarray(v) if v.ndim > 1 else v.copy() for k, v in sd.items()}
  assert sd["model.diffusion_model.input_blocks.0.0.weight"].shape == (64, 4, 3, 3)
  loaded = C.from_compvis_state_dict(sd, ucfg, TXT_CFG, acfg, with_encoder=False, kl=True)
  for part in w:
    for k in w[part]:
      assert np.array_equal(loaded[part][k], w[part][k]), (part, k)
  s = _sampler(dev, dtype, loaded)
  B = 2
  x_T = np.random.default_rng(4).standard_normal((B, 16, 16, 4)).astype(np.float32)
  ids = _ids(B, 7)
  got = s.ddim_p_sample_loop(ids, [B, 16, 16, 4], 5., x_T=x_T)
  ref = O.ddim_p_sample_loop(ids, x_T, loaded, LDM, guidance_scale=5., num_heads=8)
  r = rel(got, ref)
  print(f"CompVis-loaded weights, 10-step loop [{dtype}]: rel={r:.3e}")
  assert r < 5 * REL[dtype]


def test_autoencoder_call_round_trips(dev):
  from ldm_tf2_amd.autoencoder import AutoencoderKL, AutoencoderVQ
  m = Wt.decoder_manifest(**KL_CFG)
  m.update(Wt.encoder_manifest(**KL_CFG, image_size=64))
  w = Wt.init_weights(m, seed=2, mode="random", scope="autoencoder")
  g = torch.Generator().manual_seed(21)
  img = torch.rand(2, 64, 64, 3, generator=g) * 2 - 1
  noise = torch.randn(2, 8, 8, 4, generator=g)
  ae = AutoencoderKL(**KL_CFG, weights=w, dtype=torch.float32, device=dev)
  rec, post = ae(img, noise=noise)                       # autoencoder.py:344-351
  mom = O.encoder_forward(img, w)
  mean, _, sample = O.diagonal_gaussian(mom, noise)
  assert rel(rec, O.decoder_forward(sample, w)) < 4 * REL[torch.float32]
  rec_mode, _ = ae.call(img, sample_posterior=False)
  assert rel(rec_mode, O.decoder_forward(mean, w)) < 4 * REL[torch.float32]
  assert rel(post._moments, mom) < REL[torch.float32]
  vq = dict(latent_channels=4, channels=64, num_blocks=2, multipliers=(1, 2, 2, 4), attention_resolutions=(8,),
            vocab_size=512)
  mv = Wt.decoder_manifest(**vq, latent_size=8)
  mv.update(Wt.encoder_manifest(**vq, image_size=64, double_z=False))
  wv = Wt.init_weights(mv, seed=2, mode="random", scope="autoencoder")
  aev = AutoencoderVQ(**vq, latent_size=8, weights=wv, dtype=torch.float32, device=dev)
  out, loss, idx = aev(img, return_indices=True)         # autoencoder.py:438-444
  out2, loss2 = aev(img)
  assert torch.equal(out, out2) and tuple(out.shape) == (2, 64, 64, 3) and idx.dtype == torch.int64
  zr, qr, lossr, idxr = O.vq_encode(img, wv, attention_resolutions=(8,), beta=0.25)
  same = idx.cpu() == idxr
  assert same.float().mean().item() >= 0.98
  if bool(same.all()):
    assert rel(out, O.decoder_forward(qr, wv, attention_resolutions=(8,))) < 4 * REL[torch.float32]
  assert abs(loss.item() - lossr.item()) < 1e-3 * abs(lossr.item())


def test_vq_f8_real_configuration_decodes(dev):
  """all_in_one_config.yaml:80-89: 16384-code codebook, multipliers (1,2,2,4), attention at 32
  (the three level-3 UpBlocks attend at the 32x32 latent size, autoencoder.py:176)."""
  from ldm_tf2_amd.autoencoder import AutoencoderVQ
  cfg = dict(latent_channels=4, channels=128, num_blocks=2, multipliers=(1, 2, 2, 4), attention_resolutions=(32,),
             vocab_size=16384)
  m = Wt.decoder_manifest(**cfg, latent_size=32)
  assert sum("/attention/" in k and k.startswith("decoder/up/") for k in m) == 3 * 10
  w = Wt.init_weights(m, seed=4, mode="random", scope="autoencoder")
  g = np.random.default_rng(8)
  cb = w["quantize/kernel"]
  codes = g.integers(0, 16384, size=(1, 32, 32))
  z = cb[codes].astype(np.float32)          # exact code vectors: the nearest code is unambiguous
  zq, idx = O.vq_nearest(torch.from_numpy(z), torch.from_numpy(cb))
  assert np.array_equal(idx.numpy().reshape(codes.shape), codes)
  torch.set_num_threads(16)
  with torch.no_grad():
    ref = O.decoder_forward(zq, w, attention_resolutions=(32,))
  for dtype in (torch.float32, torch.bfloat16):
    ae = AutoencoderVQ(**cfg, latent_size=32, weights=w, dtype=dtype, device=dev)
    got = ae.decode(torch.from_numpy(z) * 0.18215, force_quantize=True, scale_factor=0.18215)   # model_runners.py:431
    assert tuple(got.shape) == (1, 256, 256, 3)
    r = rel(got, ref)
    print(f"VQ-f8 (16384 codes, attention at 32) decode [{dtype}]: rel={r:.3e}")
    assert r < REL[dtype]


def test_progressive_sampler_at_full_size_chunks_the_decode(dev):
  """B=4, record_freq=1 over 40 of the DDIM steps -> 160 frames per progress tensor: one decoder
  call on them would need a 5.4 GB (f32 10.7 GB) operand; the decoder chunks.  Run in
  bf16 with a 2-step U-Net budget: the U-Net is full size, only the number of steps is reduced."""
  from ldm_tf2_amd.autoencoder import AutoencoderKL
  from ldm_tf2_amd.model_runners import LatentDiffusionModelSampler
  from ldm_tf2_amd.transformer import TransformerModel
  from ldm_tf2_amd.unet import UNet
  ucfg = dict(model_channels=320, out_channels=4, num_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8)
  kcfg = dict(latent_channels=4, channels=128, num_blocks=2, multipliers=(1, 2, 4, 4))
  tcfg = dict(vocab_size=30522, encoder_stack_size=1, hidden_size=1280, num_heads=8, size_per_head=64,
              max_seq_len=77, filter_size=5120)
  dt = torch.bfloat16
  unet = UNet(**ucfg, dtype=dt, device=dev)
  ae = AutoencoderKL(**kcfg, dtype=dt, device=dev)
  txt = TransformerModel(**tcfg, dtype=dt, device=dev)
  ldm = dict(LDM, num_ddim_steps=40, eta=1.0)
  s = LatentDiffusionModelSampler(unet, ae, txt, verbose=False, **ldm)
  B = 4
  ids = np.concatenate([np.tile([[101, 102] + [0] * 75], (B, 1)),
                        np.tile(np.random.default_rng(1).integers(0, 30522, size=(1, 77)), (B, 1))], 0)
  images, sp, xp = s.ddim_p_sample_loop_progressive(ids, [B, 32, 32, 4], 5., record_freq=1, seed=0)
  assert tuple(sp.shape) == (B, 40, 256, 256, 3) and tuple(xp.shape) == tuple(sp.shape)
  assert bool(torch.isfinite(sp).all()) and bool(torch.isfinite(xp).all())
  # slot r holds the sample after the step with index r (model_runners.py:545-553): slot 0 is the final x_0
  # (decoded in a 16-frame chunk vs the 4-image call: other GEMM tiles, i.e. bf16 rounding order)
  assert rel(sp[:, 0], images.cpu()) < 2e-2
  # a chunked decode equals decoding the same frames on their own
  lat = torch.randn(37, 32, 32, 4, generator=torch.Generator().manual_seed(3))
  whole = ae.decode(lat, scale_factor=0.18215)
  part = ae.decode(lat[30:33], scale_factor=0.18215)
  assert rel(whole[30:33], part.cpu()) < 2e-2
  # the float32 twin of the same comparison: chunking only changes which tiles (i.e. which summation
  # order) a frame meets, so in f32 the chunked and the stand-alone decode agree to rounding
  ae32 = AutoencoderKL(**kcfg, dtype=torch.float32, device=dev)
  whole32 = ae32.decode(lat, scale_factor=0.18215)
  part32 = ae32.decode(lat[30:33], scale_factor=0.18215)
  r32 = rel(whole32[30:33], part32.cpu())
  print(f"chunked vs stand-alone decode, f32: rel={r32:.3e}")
  assert r32 < 1e-5


def test_two_samplers_on_two_streams_do_not_share_workspaces(dev):
  """Split-K slabs are per model (ops.workspace_scope): two samplers stepping concurrently on
  different streams give the results they give alone."""
  w = _weights()
  B = 2
  x_T = np.random.default_rng(3).standard_normal((B, 16, 16, 4)).astype(np.float32)
  ids = _ids(B, 1)
  alone = _sampler(dev, torch.float32, w).ddim_p_sample_loop(ids, [B, 16, 16, 4], 5., x_T=x_T).clone()
  s1, s2 = _sampler(dev, torch.float32, w), _sampler(dev, torch.float32, w)
  assert s1._unet._ws.data_ptr() != s2._unet._ws.data_ptr()
  st1, st2 = torch.cuda.Stream(), torch.cuda.Stream()
  outs = [None, None]
  for _ in range(2):
    with torch.cuda.stream(st1):
      outs[0] = s1.ddim_p_sample_loop(ids, [B, 16, 16, 4], 5., x_T=x_T)
    with torch.cuda.stream(st2):
      outs[1] = s2.ddim_p_sample_loop(ids, [B, 16, 16, 4], 5., x_T=x_T)
  torch.cuda.synchronize()
  assert torch.equal(outs[0], alone) and torch.equal(outs[1], alone)
