"""Op-level parity of every C-ABI kernel against the CPU oracle's op restatements
(oracle/ldm_oracle.py), through ctypes -> libldm_hip.so.  float32: tight tolerance
(summation order differs); bfloat16: inputs are rounded to bf16 first and the
oracle runs on the rounded values in float32, tolerance = bf16 output rounding."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ldm_oracle as O  # noqa: E402


def ops():
  from ldm_tf2_amd import ops as _ops
  return _ops


DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: dict(rtol=2e-4, atol=2e-4), torch.bfloat16: dict(rtol=2e-2, atol=2e-2)}


def rnd(shape, dtype, seed, scale=1.0):
  g = torch.Generator().manual_seed(seed)
  return (torch.randn(*shape, generator=g) * scale).to(dtype)


def close(got, ref, dtype, scale=1.0):
  got = got.detach().float().cpu()
  ref = ref.float()
  tol = TOL[dtype]
  err = (got - ref).abs().max().item()
  assert torch.allclose(got, ref, rtol=tol["rtol"], atol=tol["atol"] * scale), f"max err {err}"


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11, 12, 17, 18, 19])
@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (64, 192, 1280), (1024, 64, 64), (700, 640, 128)])
def test_linear(dev, dtype, tile, M, N, K):
  if tile >= 9 and dtype != torch.bfloat16:
    pytest.skip("tiles 9-12 are the bf16 16x16x32 MFMA path")
  x, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
  bias = rnd((N,), torch.float32, 3)
  res = rnd((M, N), dtype, 4)
  out = torch.empty(M, N, dtype=dtype, device=dev)
  ops().linear(x.to(dev), w.to(dev), out, bias=bias.to(dev), residual=res.to(dev), tile=tile)
  ref = x.float() @ w.float().t() + bias + res.float()
  close(out, ref, dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("act", ["gelu", "silu", "geglu"])
def test_linear_act_splitk(dev, dtype, act):
  o = ops()
  M, N, K = 96, 256, 2560
  x, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
  bias = rnd((N,), torch.float32, 3)
  y = x.float() @ w.float().t() + bias
  if act == "gelu":
    ref, code, nout = O.gelu(y), o.ACT_GELU, N
  elif act == "silu":
    ref, code, nout = O.silu(y), o.ACT_SILU, N
  else:
    # device layout: blocks of 64 rows = 32 value rows then their 32 gate rows
    yv = y.reshape(M, N // 64, 2, 32)
    ref, code, nout = (yv[:, :, 0] * O.gelu(yv[:, :, 1])).reshape(M, N // 2), o.ACT_GEGLU, N // 2
  for split in (1, 0, 5):
    out = torch.zeros(M, nout, dtype=dtype, device=dev)
    o.linear(x.to(dev), w.to(dev), out, bias=bias.to(dev), act=code, split_k=split)
    close(out, ref, dtype)
  # deep-ring small tiles (17-19): every activation epilogue, with and without split-K (K = 2560: 40 / 80
  # K-tiles, the ring runs full; split 5 leaves 8 / 16 per slab)
  for tile in ((19,) if act == "geglu" else (17, 18, 19)):
    for split in (1, 5):
      out = torch.zeros(M, nout, dtype=dtype, device=dev)
      o.linear(x.to(dev), w.to(dev), out, bias=bias.to(dev), act=code, split_k=split, tile=tile)
      close(out, ref, dtype)
  if dtype == torch.bfloat16:
    # 16x16x32 MFMA tiles: every activation epilogue and the split-K slab layout
    for tile in ((11, 12) if act == "geglu" else (9, 10, 11, 12)):
      for split in (1, 4):
        out = torch.zeros(M, nout, dtype=dtype, device=dev)
        o.linear(x.to(dev), w.to(dev), out, bias=bias.to(dev), act=code, split_k=split, tile=tile)
        close(out, ref, dtype)


@pytest.mark.parametrize("dtype", DT)
def test_linear_strided_f32out_addend(dev, dtype):
  o = ops()
  M, N, K = 256, 128, 192
  buf = rnd((M, 320), dtype, 1).to(dev)
  x = buf[:, 64:64 + K]
  w = rnd((N, K), dtype, 2, K ** -0.5)
  addend = rnd((4, N), torch.float32, 5)
  outbuf = torch.zeros(M, 256, dtype=torch.float32, device=dev)
  out = outbuf[:, 128:]
  o.linear(x, w.to(dev), out, addend=addend.to(dev), add_rows=64, alpha=0.5)
  ref = 0.5 * (x.float().cpu() @ w.float().t()) + addend.repeat_interleave(64, 0)
  close(out, ref, dtype)
  assert outbuf[:, :128].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cfg", [
    dict(B=2, H=16, W=16, Cin=64, Cout=128, stride=1, up=False),
    dict(B=3, H=8, W=8, Cin=128, Cout=64, stride=2, up=False),
    dict(B=2, H=8, W=8, Cin=64, Cout=64, stride=1, up=True),
    dict(B=1, H=4, W=4, Cin=256, Cout=320, stride=1, up=False),
])
def test_conv3x3(dev, dtype, cfg):
  o = ops()
  B, H, W, Cin, Cout = cfg["B"], cfg["H"], cfg["W"], cfg["Cin"], cfg["Cout"]
  x = rnd((B, H, W, Cin), dtype, 1)
  k = rnd((3, 3, Cin, Cout), dtype, 2, (9 * Cin) ** -0.5)     # HWIO
  bias = rnd((Cout,), torch.float32, 3)
  addend = rnd((B, Cout), torch.float32, 4)
  xin = O.upsample_nearest2x(x.float()) if cfg["up"] else x.float()
  ref = O.conv2d(xin, k.float(), bias, stride=cfg["stride"]) + addend[:, None, None, :]
  OH, OW = ref.shape[1], ref.shape[2]
  res = rnd((B, OH, OW, Cout), dtype, 6)
  ref = ref + res.float()
  wt = k.permute(3, 0, 1, 2).reshape(Cout, 9 * Cin).contiguous().to(dev)
  bf = (9, 10, 11, 12) if dtype == torch.bfloat16 else ()     # 16x16x32 MFMA tiles: bf16 only
  for tile in (0, 1, 2, 3, 4, 6, 7, 8) + bf + (17, 18, 19):   # implicit GEMM tiles (17-19: deep rings)
    out = torch.zeros(B, OH, OW, Cout, dtype=dtype, device=dev)
    o.conv3x3(x.to(dev), wt, out, bias=bias.to(dev), stride=cfg["stride"], upsample=cfg["up"],
              addend=addend.to(dev), residual=res.to(dev), tile=tile)
    close(out, ref, dtype)


@pytest.mark.parametrize("dtype", DT)
def test_conv3x3_channel_slices(dev, dtype):
  """input and output are channel slices of wider NHWC buffers (free skip concat)."""
  o = ops()
  B, H, W = 2, 8, 8
  wide = rnd((B, H, W, 192), dtype, 1).to(dev)
  x = wide[..., 64:192]
  k = rnd((3, 3, 128, 64), dtype, 2, (9 * 128) ** -0.5)
  wt = k.permute(3, 0, 1, 2).reshape(64, 9 * 128).contiguous().to(dev)
  obuf = torch.zeros(B, H, W, 128, dtype=dtype, device=dev)
  o.conv3x3(x, wt, obuf[..., 64:])
  ref = O.conv2d(x.float().cpu(), k.float(), None)
  close(obuf[..., 64:], ref, dtype)
  assert obuf[..., :64].float().abs().max().item() == 0


@pytest.mark.parametrize("dtype", DT)
def test_conv3x3_small(dev, dtype):
  o = ops()
  x = rnd((2, 16, 16, 4), torch.float32, 1)
  k = rnd((3, 3, 4, 64), torch.float32, 2, 0.2)
  b = rnd((64,), torch.float32, 3)
  out = torch.empty(2, 16, 16, 64, dtype=dtype, device=dev)
  o.conv3x3_small(x.to(dev), k.to(dev), b.to(dev), out)
  close(out, O.conv2d(x, k, b), dtype)
  for cout in (3, 4):
    x = rnd((2, 16, 16, 64), dtype, 4)
    k = rnd((3, 3, 64, cout), torch.float32, 5, 0.05)
    b = rnd((cout,), torch.float32, 6)
    out = torch.empty(2, 16, 16, cout, dtype=torch.float32, device=dev)
    o.conv3x3_small(x.to(dev), k.to(dev), b.to(dev), out)
    close(out, O.conv2d(x.float(), k, b), dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 16, 16, 320), (3, 4, 4, 2560), (2, 32, 32, 64), (1, 8, 8, 960)])
@pytest.mark.parametrize("silu", [False, True])
def test_groupnorm(dev, dtype, shape, silu):
  o = ops()
  B, H, W, Cc = shape
  x = rnd(shape, dtype, 1) * 2 + 0.5
  gamma, beta = rnd((Cc,), torch.float32, 2) * 0.2 + 1, rnd((Cc,), torch.float32, 3) * 0.2
  wide = torch.zeros(B, H, W, Cc + 64, dtype=dtype, device=dev)
  xs = wide[..., 64:]
  xs.copy_(x)
  out = torch.empty(shape, dtype=dtype, device=dev)
  o.groupnorm(xs, gamma.to(dev), beta.to(dev), out, eps=1e-5, silu=silu)
  ref = O.group_norm(x.float(), gamma, beta, eps=1e-5)
  if silu:
    ref = O.silu(ref)
  close(out, ref, dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(3, 32, 32, 320), (9, 16, 16, 640), (2, 8, 8, 1280), (5, 4, 4, 1280),
                                   (2, 8, 8, 2560), (2, 16, 16, 1920), (1, 32, 32, 960), (2, 32, 32, 640),
                                   (2, 64, 64, 320), (2, 5, 7, 320),
                                   # one-wave-per-slab variants (taken when the launch has >= 512 waves)
                                   (32, 8, 8, 1280), (32, 4, 4, 2560), (32, 8, 8, 1920), (32, 16, 16, 1280),
                                   (33, 8, 8, 640)])
def test_groupnorm_single_launch(dev, dtype, shape):
  """The register-resident single-launch kernel (every U-Net GroupNorm shape) against the
  oracle, against the two-launch path, and run-to-run identical (fixed-order reductions)."""
  o = ops()
  from ldm_tf2_amd._lib import lib
  B, H, W, Cc = shape
  x = rnd(shape, dtype, 1) * 2 + 0.5
  gamma, beta = rnd((Cc,), torch.float32, 2) * 0.2 + 1, rnd((Cc,), torch.float32, 3) * 0.2
  wide = torch.zeros(B, H, W, Cc + 64, dtype=dtype, device=dev)
  xs = wide[..., 64:]
  xs.copy_(x)
  sup = lib.ldm_groupnorm_fused_supported(B, H * W, Cc, 32, o.code(dtype))
  if shape != (2, 64, 64, 320):     # 4096 pixels x 5 chunks is beyond the register slab: two launches
    assert sup == 1, shape
  ref = O.silu(O.group_norm(x.float(), gamma, beta, eps=1e-6))
  outs = []
  for fused in (True, True, False):
    out = torch.full(shape, float("nan"), dtype=dtype, device=dev)
    o.groupnorm(xs, gamma.to(dev), beta.to(dev), out, eps=1e-6, silu=True, fused=fused)
    torch.cuda.synchronize()
    close(out, ref, dtype, scale=2.0)
    outs.append(out)
  assert torch.equal(outs[0], outs[1])
  if dtype == torch.float32:
    assert (outs[0] - outs[2]).abs().max().item() < 2e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Cc", [64, 320, 1280])
def test_layernorm(dev, dtype, Cc):
  o = ops()
  x = rnd((77, Cc), dtype, 1) * 3 + 1
  gamma, beta = rnd((Cc,), torch.float32, 2) * 0.2 + 1, rnd((Cc,), torch.float32, 3) * 0.2
  out = torch.empty(77, Cc, dtype=dtype, device=dev)
  o.layernorm(x.to(dev), gamma.to(dev), beta.to(dev), out)
  close(out, O.layer_norm(x.float(), gamma, beta), dtype, scale=2.0)


def test_softmax_rows(dev):
  o = ops()
  x = rnd((50, 1000), torch.float32, 1) * 4
  ref = torch.softmax(x * 0.3, dim=-1)
  xd = x.to(dev)
  outb = torch.empty(50, 1000, dtype=torch.bfloat16, device=dev)
  o.softmax_rows(xd, outb, 0.3)
  close(outb, ref, torch.bfloat16)
  o.softmax_rows(xd, xd, 0.3)
  close(xd, ref, torch.float32)


def _attn_case(dev, dtype, R, H, S, Tq, Tk, sp):
  o = ops()
  q = rnd((R, Tq, H, S), dtype, 1)
  k = rnd((R, Tk, H, S), dtype, 2)
  v = rnd((R, Tk, H, S), dtype, 3)
  scale = S ** -0.5
  logits = torch.einsum("nqhs,nchs->nhqc", q.float(), k.float()) * scale
  ref = torch.einsum("nhqc,nchs->nqhs", torch.softmax(logits, dim=3), v.float())
  pad = lambda t: torch.nn.functional.pad(t, (0, sp - S))
  qd = pad(q).reshape(R, Tq, H * sp).to(dev)
  kd = pad(k).reshape(R, Tk, H * sp).to(dev)
  # the padding columns [Tk, ldvt) of V^T are NaN: the kernel masks them (include/ldm_hip.h), a
  # caller need not initialise them
  tkp = (Tk + 7) // 8 * 8 + 8
  vt = torch.full((R, H * sp, tkp), float("nan"), dtype=dtype)
  vt[:, :, :Tk] = pad(v).reshape(R, Tk, H * sp).permute(0, 2, 1)
  out = torch.full((R, Tq, H * sp), float("nan"), dtype=dtype, device=dev)
  o.attention(qd, kd, vt.to(dev), out, H, sp, scale)
  got = out.reshape(R, Tq, H, sp)
  close(got[..., :S], ref, dtype)
  assert got[..., S:].float().abs().max().item() == 0 if sp > S else True


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [
    dict(R=2, H=8, S=40, Tq=256, Tk=256, sp=64),
    dict(R=2, H=8, S=40, Tq=256, Tk=77, sp=64),
    dict(R=1, H=4, S=80, Tq=64, Tk=64, sp=96),
    dict(R=2, H=2, S=160, Tq=16, Tk=16, sp=160),
    dict(R=1, H=2, S=160, Tq=64, Tk=77, sp=160),
    dict(R=2, H=8, S=64, Tq=77, Tk=77, sp=64),
    dict(R=1, H=8, S=8, Tq=200, Tk=130, sp=32),
    dict(R=2, H=8, S=40, Tq=256, Tk=256, sp=48),     # head sizes the U-Net uses: 40->48, 80->80
    dict(R=2, H=8, S=40, Tq=192, Tk=77, sp=48),
    dict(R=1, H=8, S=80, Tq=64, Tk=200, sp=80),
    dict(R=1, H=4, S=40, Tq=1000, Tk=333, sp=48),    # 8-wave workgroups (Tq >= 256), ragged queries and keys, 6 tiles
    dict(R=1, H=2, S=40, Tq=256, Tk=64, sp=48),      # 8-wave workgroups, ONE tile
])
def test_attention(dev, dtype, case):
  _attn_case(dev, dtype, **case)


def test_attention_spike(dev):
  """forces the online-softmax rescale: one key dominates late in the sequence."""
  o = ops()
  R, H, S, T, sp = 1, 1, 64, 256, 64
  q = rnd((R, T, H, S), torch.float32, 1)
  k = rnd((R, T, H, S), torch.float32, 2)
  v = rnd((R, T, H, S), torch.float32, 3)
  k[0, 200, 0] = q[0, 5, 0] * 4
  scale = S ** -0.5
  logits = torch.einsum("nqhs,nchs->nhqc", q, k) * scale
  ref = torch.einsum("nhqc,nchs->nqhs", torch.softmax(logits, dim=3), v)
  out = torch.empty(R, T, H * sp, device=dev)
  o.attention(q.reshape(R, T, S).to(dev), k.reshape(R, T, S).to(dev),
              v.reshape(R, T, S).permute(0, 2, 1).contiguous().to(dev), out, H, sp, scale)
  close(out.reshape(R, T, H, sp), ref, torch.float32)


@pytest.mark.parametrize("dtype", DT)
def test_bmm_nt_and_transposed(dev, dtype):
  o = ops()
  a, w = rnd((3, 100, 128), dtype, 1), rnd((3, 80, 128), dtype, 2, 0.1)
  out = torch.empty(3, 100, 80, dtype=torch.float32, device=dev)
  o.bmm_nt(a.to(dev), w.to(dev), out, alpha=0.25)
  close(out, 0.25 * torch.einsum("bmk,bnk->bmn", a.float(), w.float()), dtype)
  ws = rnd((96, 128), dtype, 3, 0.1)
  outT = torch.zeros(3, 96, 104, dtype=dtype, device=dev)
  o.bmm_nt(a.to(dev), ws.to(dev), outT, transposed_out=True)
  ref = torch.einsum("bmk,nk->bnm", a.float(), ws.float())
  close(outT[:, :, :100], ref, dtype)
  assert outT[:, :, 100:].float().abs().max().item() == 0


def test_time_embedding_gemv(dev):
  o = ops()
  t = torch.tensor([981, 1, 500], dtype=torch.int32)
  out = torch.empty(3, 320, device=dev)
  o.time_embedding(out, 320, t_rows=t.to(dev))
  ref = O.get_time_embedding(t.numpy(), 320)
  assert torch.allclose(out.cpu(), ref, atol=2e-4, rtol=0)
  steps = torch.arange(1, 1000, 20, dtype=torch.int32, device=dev)
  idx = torch.tensor([49], dtype=torch.int32, device=dev)
  o.time_embedding(out, 320, steps=steps, index=idx)
  assert torch.allclose(out.cpu(), ref[0:1].expand(3, -1), atol=2e-4, rtol=0)
  for dtype in DT:
    x = rnd((3, 320), torch.float32, 1)
    w = rnd((1280, 320), dtype, 2, 0.05)
    b = rnd((1280,), torch.float32, 3)
    y = torch.empty(3, 1280, device=dev)
    o.gemv(x.to(dev), w.to(dev), b.to(dev), y, act_in=o.ACT_SILU, act_out=o.ACT_SILU)
    close(y, O.silu(O.silu(x) @ w.float().t() + b), torch.float32)


def test_cfg_ddim_update(dev):
  o = ops()
  sched = O.make_schedule(1000, 0.00085, 0.012, 0.7, 50)
  coef = np.stack([sched["ddim_sqrt_recip_alphas_cumprod"], sched["ddim_sqrt_recipm1_alphas_cumprod"],
                   sched["ddim_alphas_cumprod_prev"], sched["ddim_sigmas"]], 1).astype(np.float32)
  B = 3
  eps = rnd((2 * B, 8, 8, 4), torch.float32, 1)
  xt = rnd((B, 8, 8, 4), torch.float32, 2)
  noise = rnd((B, 8, 8, 4), torch.float32, 3)
  idx = torch.tensor([17], dtype=torch.int32, device=dev)
  xo = torch.empty(B, 8, 8, 4, device=dev)
  xu = torch.empty(2 * B, 8, 8, 4, dtype=torch.bfloat16, device=dev)
  o.cfg_ddim_update(eps.to(dev), xt.to(dev), xo, torch.from_numpy(coef).to(dev), idx, 5.0,
                    noise=noise.to(dev), x_unet_out=xu, dec_index=True)
  ref, _ = O.ddim_update(xt, eps[:B], eps[B:], sched, 17, 5.0, noise)
  assert torch.allclose(xo.cpu(), ref, rtol=1e-5, atol=1e-5)
  assert idx.item() == 16
  assert torch.equal(xu[:B].cpu(), ref.to(torch.bfloat16)) or torch.allclose(
      xu[:B].float().cpu(), ref, rtol=1e-2, atol=1e-2)
  assert torch.equal(xu[:B], xu[B:])


def test_post_quant_vq_embedding_minmax_cast(dev):
  o = ops()
  z = rnd((2, 8, 8, 4), torch.float32, 1)
  k, b = rnd((4, 4), torch.float32, 2), rnd((4,), torch.float32, 3)
  out = torch.empty(2, 8, 8, 4, device=dev)
  o.post_quant(z.to(dev), 0.18215, k.to(dev), b.to(dev), out)
  assert torch.allclose(out.cpu(), O.dense(z / 0.18215, k, b), rtol=1e-5, atol=1e-5)
  cb = rnd((1000, 4), torch.float32, 4)
  q = torch.empty(2, 8, 8, 4, device=dev)
  ind = torch.empty(128, dtype=torch.int64, device=dev)
  o.vq_nearest(z.to(dev), cb.to(dev), q, ind)
  qr, ir = O.vq_nearest(z, cb)
  assert torch.equal(ind.cpu(), ir)
  assert torch.allclose(q.cpu(), qr, rtol=1e-6, atol=1e-6)
  ids = torch.randint(0, 500, (3, 77), generator=torch.Generator().manual_seed(0))
  tok, pos = rnd((500, 64), torch.float32, 5), rnd((77, 64), torch.float32, 6)
  e = torch.empty(3, 77, 64, device=dev)
  o.embedding(ids.to(dev), tok.to(dev), pos.to(dev), e)
  assert torch.equal(e.cpu(), tok[ids] + pos[None])
  img = rnd((3, 32, 32, 3), torch.float32, 7)
  u8 = torch.empty(3, 32, 32, 3, dtype=torch.uint8, device=dev)
  o.minmax_u8(img.to(dev), u8)
  # byte work is bit-exact (run_ldm_sampler.py:18-25): identical f32 input -> identical bytes.  The
  # kernel follows the reference literally ((x - min) / (max - min), * 255, truncate) with an IEEE
  # divide and no contraction across the multiply, so there is nothing to tolerate.
  ref = O.tensor_to_image(img.numpy())
  assert np.array_equal(u8.cpu().numpy(), ref)
  # bf16 images (the bf16 decoder's output dtype): the oracle is fed the same rounded values
  imgb = img.to(torch.bfloat16)
  o.minmax_u8(imgb.to(dev), u8)
  assert np.array_equal(u8.cpu().numpy(), O.tensor_to_image(imgb.float().numpy()))
  # a larger, image-sized case with a wide dynamic range (every byte value is hit many times)
  big = (rnd((2, 256, 256, 3), torch.float32, 8) * 3.0).contiguous()
  u8b = torch.empty(2, 256, 256, 3, dtype=torch.uint8, device=dev)
  o.minmax_u8(big.to(dev), u8b)
  assert np.array_equal(u8b.cpu().numpy(), O.tensor_to_image(big.numpy()))
  c = torch.empty(3, 32, 32, 3, dtype=torch.bfloat16, device=dev)
  o.cast(img.to(dev), c)
  assert torch.equal(c.cpu(), img.to(torch.bfloat16))


def test_error_reporting(dev):
  from ldm_tf2_amd._lib import LdmHipError
  o = ops()
  x = torch.zeros(8, 100, device=dev)          # K=100 ok for f32, Cin check fails for conv
  with pytest.raises(LdmHipError):
    o.conv3x3(torch.zeros(1, 4, 4, 20, device=dev), torch.zeros(8, 180, device=dev),
              torch.zeros(1, 4, 4, 8, device=dev))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,K", [(300, 320), (1024, 128), (77, 1280)])
def test_linear_with_layernorm_output(dev, dtype, M, K):
  """ldm_gemm's second output: LayerNorm (unet.py:309-313) of the row it has just stored."""
  o = ops()
  N = 320
  assert o.linear_ln_supported(N, dtype) and not o.linear_ln_supported(640, dtype)
  x, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
  bias, res = rnd((N,), torch.float32, 3), rnd((M, N), dtype, 4) * 2 + 0.7
  gamma, beta = rnd((N,), torch.float32, 5) * 0.2 + 1, rnd((N,), torch.float32, 6) * 0.2
  out = torch.full((M, N), float("nan"), dtype=dtype, device=dev)
  lnb = torch.full((M, N + 64), float("nan"), dtype=dtype, device=dev)
  ln_out = lnb[:, 64:]                                     # strided second output
  o.linear(x.to(dev), w.to(dev), out, bias=bias.to(dev), residual=res.to(dev),
           ln=(gamma.to(dev), beta.to(dev), ln_out, 1e-5))
  torch.cuda.synchronize()
  ref = x.float() @ w.float().t() + bias + res.float()
  close(out, ref, dtype)
  # the LayerNorm is of the row AS STORED: identical to a separate ldm_layernorm of `out`
  sep = torch.empty(M, N, dtype=dtype, device=dev)
  o.layernorm(out, gamma.to(dev), beta.to(dev), sep, 1e-5)
  torch.cuda.synchronize()
  close(ln_out, O.layer_norm(out.float().cpu(), gamma, beta), dtype, scale=2.0)
  assert (ln_out.float() - sep.float()).abs().max().item() <= (2e-5 if dtype == torch.float32 else 4e-2)
  assert torch.isnan(lnb[:, :64].float()).all()
  from ldm_tf2_amd._lib import LdmHipError
  with pytest.raises(LdmHipError):                         # a tile cannot hold a 640-wide row: loud error
    o.linear(x.to(dev), rnd((640, K), dtype, 2).to(dev), torch.empty(M, 640, dtype=dtype, device=dev),
             ln=(torch.ones(640, device=dev), torch.zeros(640, device=dev), torch.empty(M, 640, dtype=dtype, device=dev), 1e-5))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("R,T,hs,K,tile", [(3, 64, 128, 128, 0), (2, 256, 640, 640, 0), (2, 1024, 384, 320, 0),
                                          (2, 256, 640, 640, 6), (5, 16, 128, 192, 3)])
def test_linear_with_transposed_second_output(dev, dtype, R, T, hs, K, tile):
  """q|k|v in one launch: q|k row-major, v straight into the attention kernel's V^T [R, hs, T'] (ldm_gemm out2)."""
  o = ops()
  x = rnd((R, T, K), dtype, 1)
  w = rnd((3 * hs, K), dtype, 2, K ** -0.5)
  tp = (T + 7) // 8 * 8
  qk = torch.full((R, T, 2 * hs), float("nan"), dtype=dtype, device=dev)
  vt = torch.zeros(R, hs, tp, dtype=dtype, device=dev)
  o.linear(x.to(dev), w.to(dev), qk, out2=vt, tile=tile)
  torch.cuda.synchronize()
  ref = x.float() @ w.float().t()
  close(qk, ref[..., :2 * hs], dtype)
  close(vt[..., :T], ref[..., 2 * hs:].transpose(1, 2), dtype)
  assert vt[..., T:].float().abs().max().item() == 0 if tp > T else True
  # identical to the two separate launches it replaces
  qk2 = torch.empty_like(qk)
  vt2 = torch.zeros_like(vt)
  o.linear(x.to(dev), w[:2 * hs].contiguous().to(dev), qk2, tile=tile)
  o.bmm_nt(x.to(dev), w[2 * hs:].contiguous().to(dev), vt2, transposed_out=True)
  torch.cuda.synchronize()
  assert (qk.float() - qk2.float()).abs().max().item() <= (1e-5 if dtype == torch.float32 else 2e-2)
  assert (vt.float() - vt2.float()).abs().max().item() <= (1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 32, 4, 320), (1, 9, 7, 3, 128), (2, 8, 8, 4, 12), (1, 16, 16, 4, 512)])
def test_conv3x3_small_in_paths(dev, dtype, B, H, W, Cin, Cout):
  """conv_in: LDS-resident-weights kernel (Cout % 8 == 0, kernel <= 64 KB) and the per-thread-weights
  fallback (Cout = 12; Cout = 512 is 72 KB) against the oracle, writing into a channel slice."""
  o = ops()
  x = rnd((B, H, W, Cin), torch.float32, 1)
  k = rnd((3, 3, Cin, Cout), torch.float32, 2, 0.3)
  b = rnd((Cout,), torch.float32, 3)
  wide = torch.full((B, H, W, Cout + 64), 7.0, dtype=dtype, device=dev)
  out = wide[..., 64:]
  o.conv3x3_small(x.to(dev), k.to(dev), b.to(dev), out)
  torch.cuda.synchronize()
  close(out, O.conv2d(x, k, b), dtype)
  assert (wide[..., :64].float() == 7.0).all()


# ---- halo-staged stride-1 convolution (tiles 15 = 256x160, 16 = 256x128; gemm_kernel.h MODE 3) -------
@pytest.mark.parametrize("cfg", [
    dict(B=2, H=32, W=32, Cin=128, Cout=320),      # 4 M-tiles per image (8 lines each), 2 channel chunks
    dict(B=3, H=16, W=16, Cin=192, Cout=640),      # one image per M-tile, 3 chunks (odd: both patch buffers end the loop)
    dict(B=1, H=32, W=32, Cin=64, Cout=128),       # ONE chunk: the prologue patch only
    dict(B=2, H=16, W=16, Cin=320, Cout=160),      # 5 chunks
    dict(B=2, H=32, W=16, Cin=64, Cout=128),       # non-square: two 16-line tiles per image
    dict(B=4, H=8, W=32, Cin=128, Cout=160),       # an 8-line image = exactly one tile
])
def test_conv3x3_halo_ring(dev, cfg):
  """Every tap reads its A fragments from the staged (lines + 2) x (W + 2) patch at a row shift: image
  borders (zero padding comes from the staging range check), tile borders inside an image (halo lines
  of the neighbouring tile), bias + per-sample addend + residual epilogue, split-K at chunk borders."""
  o = ops()
  dtype = torch.bfloat16
  B, H, W, Cin, Cout = cfg["B"], cfg["H"], cfg["W"], cfg["Cin"], cfg["Cout"]
  x = rnd((B, H, W, Cin), dtype, 1)
  k = rnd((3, 3, Cin, Cout), dtype, 2, (9 * Cin) ** -0.5)
  bias = rnd((Cout,), torch.float32, 3)
  addend = rnd((B, Cout), torch.float32, 4)
  res = rnd((B, H, W, Cout), dtype, 6)
  ref = O.conv2d(x.float(), k.float(), bias) + addend[:, None, None, :] + res.float()
  wt = k.permute(3, 0, 1, 2).reshape(Cout, 9 * Cin).contiguous().to(dev)
  ran = 0
  for tile in (15, 16):
    if Cout % (160 if tile == 15 else 128):
      continue
    for split in (1, 2, 3):
      if split > Cin // 64:
        continue
      out = torch.zeros(B, H, W, Cout, dtype=dtype, device=dev)
      o.conv3x3(x.to(dev), wt, out, bias=bias.to(dev), addend=addend.to(dev), residual=res.to(dev), tile=tile,
                split_k=split)
      close(out, ref, dtype)
      ran += 1
  assert ran
  # a channel slice of a wider buffer as input (pixel stride > Cin), as the U-Net's concat buffers are
  wide = rnd((B, H, W, Cin + 64), dtype, 7).to(dev)
  xs = wide[..., 64:]
  ref2 = O.conv2d(xs.float().cpu(), k.float(), bias)
  tile = 15 if Cout % 160 == 0 else 16
  out = torch.zeros(B, H, W, Cout, dtype=dtype, device=dev)
  o.conv3x3(xs, wt, out, bias=bias.to(dev), tile=tile)
  close(out, ref2, dtype)
  from ldm_tf2_amd._lib import LdmHipError
  with pytest.raises(LdmHipError):                                       # stride 2: not a halo-ring problem
    o.conv3x3(x.to(dev), wt, torch.zeros(B, H // 2, W // 2, Cout, dtype=dtype, device=dev), stride=2, tile=tile)
  with pytest.raises(LdmHipError):                                       # f32: bf16 tiles only
    o.conv3x3(x.float().to(dev), wt.float(), torch.zeros(B, H, W, Cout, device=dev), tile=tile)


# ---- persistent ping-pong kernel (tiles 13 = 256x160, 14 = 256x128; gemm3_kernel.h) ---------------
@pytest.mark.parametrize("tile,N", [(13, 320), (13, 1280), (14, 384), (14, 1152)])
@pytest.mark.parametrize("M,K,nsplit", [(300, 320, 0), (1000, 384, 1), (777, 64, 2), (4096, 640, 0)])
def test_persistent_linear(dev, tile, N, M, K, nsplit):
  """Several n-tiles per workgroup (nsplit = 1: ALL n-tiles of a panel in one workgroup), ragged M,
  one-K-tile problems (every step is a tile boundary); epilogues: none, bias, bias + residual."""
  o = ops()
  dtype = torch.bfloat16
  x, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
  bias = rnd((N,), torch.float32, 3)
  res = rnd((M, N), dtype, 4)
  y = x.float() @ w.float().t()
  for b, r in ((None, None), (bias, None), (bias, res), (None, res)):
    out = torch.full((M + 3, N), 7.0, dtype=dtype, device=dev)          # rows beyond M must stay untouched
    o.linear(x.to(dev), w.to(dev), out[:M], bias=None if b is None else b.to(dev),
             residual=None if r is None else r.to(dev), tile=tile, split_k=-nsplit)
    ref = y + (0 if b is None else b) + (0 if r is None else r.float())
    close(out[:M], ref, dtype)
    assert bool((out[M:] == 7.0).all())


@pytest.mark.parametrize("act", ["gelu", "silu", "geglu"])
def test_persistent_linear_activations(dev, act):
  o = ops()
  dtype = torch.bfloat16
  M, N, K = 520, 1280, 320
  x, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
  bias = rnd((N,), torch.float32, 3)
  y = x.float() @ w.float().t() + bias
  code = {"gelu": o.ACT_GELU, "silu": o.ACT_SILU, "geglu": o.ACT_GEGLU}[act]
  if act == "geglu":
    yv = y.reshape(M, N // 64, 2, 32)
    ref, nout = (yv[:, :, 0] * O.gelu(yv[:, :, 1])).reshape(M, N // 2), N // 2
  else:
    ref, nout = {"gelu": O.gelu, "silu": O.silu}[act](y), N
  for tile in ((14,) if act == "geglu" else (13, 14)):
    for nsplit in (0, 1, 3):
      out = torch.zeros(M, nout, dtype=dtype, device=dev)
      o.linear(x.to(dev), w.to(dev), out, bias=bias.to(dev), act=code, tile=tile, split_k=-nsplit)
      close(out, ref, dtype)
  from ldm_tf2_amd._lib import LdmHipError
  with pytest.raises(LdmHipError):                                       # N not a whole number of n-tiles
    o.linear(x.to(dev), w[:200].contiguous().to(dev), torch.zeros(M, 200, dtype=dtype, device=dev), tile=13)
  with pytest.raises(LdmHipError):                                       # bf16 only
    o.linear(x.float().to(dev), w.float().to(dev), torch.zeros(M, N, device=dev), tile=13)
  with pytest.raises(LdmHipError):                                       # epilogue combination not built: loud
    o.linear(x.to(dev), w.to(dev), torch.zeros(M, N, dtype=dtype, device=dev), bias=bias.to(dev), act=o.ACT_GELU,
             residual=torch.zeros(M, N, dtype=dtype, device=dev), tile=13)


@pytest.mark.parametrize("cfg", [
    dict(B=3, H=16, W=16, Cin=128, Cout=320, stride=1, up=False),
    dict(B=2, H=32, W=32, Cin=64, Cout=128, stride=1, up=False),
    dict(B=5, H=8, W=8, Cin=192, Cout=640, stride=1, up=False),          # ragged M (320 rows), 27 K-tiles
    dict(B=2, H=16, W=16, Cin=64, Cout=160, stride=2, up=False),
    dict(B=2, H=8, W=8, Cin=128, Cout=256, stride=1, up=True),
])
def test_persistent_conv3x3(dev, cfg):
  o = ops()
  dtype = torch.bfloat16
  B, H, W, Cin, Cout = cfg["B"], cfg["H"], cfg["W"], cfg["Cin"], cfg["Cout"]
  x = rnd((B, H, W, Cin), dtype, 1)
  k = rnd((3, 3, Cin, Cout), dtype, 2, (9 * Cin) ** -0.5)
  bias = rnd((Cout,), torch.float32, 3)
  addend = rnd((B, Cout), torch.float32, 4)
  xin = O.upsample_nearest2x(x.float()) if cfg["up"] else x.float()
  base = O.conv2d(xin, k.float(), bias, stride=cfg["stride"])
  OH, OW = base.shape[1], base.shape[2]
  res = rnd((B, OH, OW, Cout), dtype, 6)
  wt = k.permute(3, 0, 1, 2).reshape(Cout, 9 * Cin).contiguous().to(dev)
  for tile in (13, 14):
    if Cout % (160 if tile == 13 else 128):
      continue
    for nsplit in (0, 1):
      # the three conv epilogues of the U-Net: bias (down / up), bias + temb addend (conv1), bias + residual (conv2)
      for ad, rs in ((None, None), (addend, None), (None, res)):
        out = torch.zeros(B, OH, OW, Cout, dtype=dtype, device=dev)
        o.conv3x3(x.to(dev), wt, out, bias=bias.to(dev), stride=cfg["stride"], upsample=cfg["up"],
                  addend=None if ad is None else ad.to(dev), residual=None if rs is None else rs.to(dev),
                  tile=tile, split_k=-nsplit)
        ref = base + (0 if ad is None else ad[:, None, None, :]) + (0 if rs is None else rs.float())
        close(out, ref, dtype)


@pytest.mark.parametrize("R,T,K,N", [(3, 64, 320, 384), (2, 1024, 320, 384), (4, 256, 640, 640), (2, 96, 128, 160)])
def test_persistent_linear_transposed(dev, R, T, K, N):
  """ops.linear_t: the product stored transposed per group of T rows (V^T of the self-attention)."""
  o = ops()
  dtype = torch.bfloat16
  x = rnd((R, T, K), dtype, 1)
  w = rnd((N, K), dtype, 2, K ** -0.5)
  tp = (T + 7) // 8 * 8 + 8                       # a padded row pitch, as the attention V^T buffers have
  out = torch.full((R, N, tp), 3.0, dtype=dtype, device=dev)
  assert o.linear_t_supported(x.to(dev), w.to(dev), out)
  o.linear_t(x.to(dev), w.to(dev), out)
  ref = torch.einsum("rtk,nk->rnt", x.float(), w.float())
  close(out[:, :, :T], ref, dtype)
  assert bool((out[:, :, T:] == 3.0).all())     # the pad columns are not touched
  # same result as the batched transposed-output GEMM the non-persistent path uses
  out2 = torch.zeros(R, N, tp, dtype=dtype, device=dev)
  o.bmm_nt(x.to(dev), w.to(dev), out2, transposed_out=True)
  close(out[:, :, :T], out2[:, :, :T].cpu(), dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,T", [(2, 64), (1, 1024), (3, 96), (1, 200)])
def test_attention_wide_head(dev, dtype, B, T):
  """Single-head attention with head dim 512 (autoencoder.py:74-97): the head dim is split over the
  four waves of a workgroup; ragged key counts are masked in the last tile; a spiked key forces the
  lazy-rescale branch."""
  o = ops()
  C = 512
  q = rnd((B, T, C), dtype, 1) * 0.7
  k = rnd((B, T, C), dtype, 2) * 0.7
  v = rnd((B, T, C), dtype, 3)
  k[0, T // 2] = q[0, 5] * 6.0                       # one key that dominates query 5 late in the sweep
  scale = C ** -0.5
  ref = torch.softmax(torch.einsum("bqc,bkc->bqk", q.float(), k.float()) * scale, -1) @ v.float()
  ldvt = (T + 7) // 8 * 8
  vt = torch.full((B, C, ldvt), float("nan"), dtype=dtype)     # NaN padding columns: masked in the kernel
  vt[:, :, :T] = v.transpose(1, 2)
  out = torch.zeros(B, T, C, dtype=dtype, device=dev)
  o.attention(q.to(dev), k.to(dev), vt.to(dev), out, 1, 512, scale)
  close(out, ref, dtype, scale=2.0)
