"""Round-3 kernels and launch forms against the CPU oracle (through the C ABI).

* LayerNorm folded into the consuming projection (ldm_gemm `ln_cs`, unet.py:309-313): the oracle
  runs layer_norm -> dense on the same bf16-rounded rows with the UNFOLDED float32 weights.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ldm_tf2_amd import layout as L  # noqa: E402
from ldm_tf2_amd import weights as Wt  # noqa: E402
from oracle import ldm_oracle as O  # noqa: E402

BF = torch.bfloat16


def ops():
  from ldm_tf2_amd import ops as _ops
  return _ops


def rnd(shape, seed, scale=1.0):
  g = torch.Generator().manual_seed(seed)
  return torch.randn(*shape, generator=g) * scale


def rel(got, ref):
  got, ref = got.detach().float().cpu().double(), ref.detach().double()
  return ((got - ref).norm() / ref.norm()).item()


def _ln_case(M, K, seed, mean_shift):
  """Rows with a per-row offset (|mean| up to `mean_shift` sigma: the fold subtracts mean * colsum from
  the accumulated product, so a large mean is the hard case) and per-row scale."""
  x = rnd((M, K), seed) * (0.5 + torch.rand(M, 1, generator=torch.Generator().manual_seed(seed + 1)) * 2.0)
  x = x + rnd((M, 1), seed + 2) * mean_shift
  gamma = 1.0 + 0.3 * rnd((K,), seed + 3)
  beta = 0.2 * rnd((K,), seed + 4)
  return x.to(BF), gamma, beta


@pytest.mark.parametrize("M,K,N,tile", [
    (2048, 320, 768, 0),       # q|k at the 32x32 level (N % 128 == 0 -> tile 14)
    (1000, 320, 768, 14),      # ragged last panel
    (1024, 1280, 1280, 13),    # 160-column n-tiles, 20 K-tiles
    (512, 640, 640, 0),        # N % 160 == 0 and N % 128 == 0
    (768, 64, 256, 0),         # ONE K-tile
])
@pytest.mark.parametrize("mean_shift", [0.0, 4.0])
def test_linear_layernorm_fold(dev, M, K, N, tile, mean_shift):
  o = ops()
  x, gamma, beta = _ln_case(M, K, 10, mean_shift)
  w = rnd((N, K), 20, K ** -0.5)
  bias = rnd((N,), 21)
  ref = O.dense(O.layer_norm(x.float(), gamma, beta, eps=1e-5), w.t(), bias)
  wq, cs, bb = L.ln_fold(w, gamma.numpy(), beta.numpy(), bias.numpy(), BF, dev)
  out = torch.full((M, N), float("nan"), dtype=BF, device=dev)
  o.linear(x.to(dev), wq, out, bias=bb, ln_fold=(cs, 1e-5), tile=tile)
  r = rel(out, ref)
  # the unfused form on the same inputs (LayerNorm kernel -> bf16 rows -> GEMM) for comparison
  ln = torch.empty(M, K, dtype=BF, device=dev)
  o.layernorm(x.to(dev), gamma.to(dev), beta.to(dev), ln, 1e-5)
  out2 = torch.empty(M, N, dtype=BF, device=dev)
  o.linear(ln, w.to(BF).to(dev), out2, bias=bias.to(dev))
  r2 = rel(out2, ref)
  print(f"LN fold M={M} K={K} N={N} shift={mean_shift}: rel {r:.3e} (LayerNorm kernel + GEMM: {r2:.3e})")
  # bf16 output rounding is 2^-9 relative per element (rel-L2 ~ 2.3e-3); the fold skips the bf16
  # rounding of the normalised rows, so it is at least as close as the two-launch form
  assert r < 4e-3 and r <= r2 * 1.25 + 1e-4


def test_linear_layernorm_fold_geglu(dev):
  """LayerNorm -> GEGLU projection (unet.py:313, :323-325) in one launch."""
  o = ops()
  M, C = 1024, 320
  x, gamma, beta = _ln_case(M, C, 30, 2.0)
  k_io = rnd((C, 8 * C), 31, C ** -0.5).numpy()
  b = rnd((8 * C,), 32).numpy()
  y = O.dense(O.layer_norm(x.float(), gamma, beta, eps=1e-5), torch.from_numpy(k_io), torch.from_numpy(b))
  ref = y[:, :4 * C] * O.gelu(y[:, 4 * C:])                       # value first, gate second
  gw, gb = L.geglu_kernel(k_io, b, torch.float32, "cpu")
  wq, cs, bb = L.ln_fold(gw, gamma.numpy(), beta.numpy(), gb.numpy(), BF, dev)
  out = torch.full((M, 4 * C), float("nan"), dtype=BF, device=dev)
  o.linear(x.to(dev), wq, out, bias=bb, act=o.ACT_GEGLU, ln_fold=(cs, 1e-5))
  r = rel(out, ref)
  print(f"LN fold + GEGLU: rel {r:.3e}")
  assert r < 5e-3


@pytest.mark.parametrize("R,T,K,N", [(2, 1024, 320, 384), (4, 256, 640, 640), (3, 64, 1280, 1280)])
def test_linear_t_layernorm_fold(dev, R, T, K, N):
  """LayerNorm -> V projection stored transposed per sample (the attention kernel's V^T layout)."""
  o = ops()
  x, gamma, beta = _ln_case(R * T, K, 40, 3.0)
  w = rnd((N, K), 41, K ** -0.5)
  ref = O.dense(O.layer_norm(x.float(), gamma, beta, eps=1e-5), w.t(), None)      # [R*T, N]
  ref = ref.reshape(R, T, N).permute(0, 2, 1)
  wq, cs, bb = L.ln_fold(w, gamma.numpy(), beta.numpy(), None, BF, dev)
  tp = T + 8
  out = torch.full((R, N, tp), 3.0, dtype=BF, device=dev)
  o.linear_t(x.reshape(R, T, K).to(dev), wq, out, bias=bb, ln_fold=(cs, 1e-5))
  r = rel(out[:, :, :T], ref)
  print(f"LN fold, transposed store R={R} T={T} K={K} N={N}: rel {r:.3e}")
  assert r < 4e-3
  assert bool((out[:, :, T:] == 3.0).all())                      # the pad columns are not touched


def test_unet_with_folded_layernorm_matches_the_oracle_and_the_unfolded_unet(dev):
  """A U-Net whose every transformer block takes the fold (fold_min_rows=1) against the oracle, and
  against the same U-Net with separate LayerNorm launches."""
  from ldm_tf2_amd.unet import UNet
  cfg = dict(model_channels=128, out_channels=4, num_blocks=1, channel_mult=(1, 2), num_heads=2)   # heads 64 and 128 -> 160 wide
  ctx_dim = 128
  w = Wt.init_weights(Wt.unet_manifest(context_dim=ctx_dim, **cfg), seed=5, mode="random", scope="unet")
  g = np.random.default_rng(6)
  x = g.standard_normal((2, 32, 32, 4)).astype(np.float32)
  ctx = g.standard_normal((2, 77, ctx_dim)).astype(np.float32)
  t = np.array([981, 21], dtype=np.int32)
  ref = O.unet_forward(x, t, ctx, w, num_heads=2)
  outs = {}
  for fold in (True, False):
    unet = UNet(**cfg, weights=w, dtype=BF, device=dev, context_dim=ctx_dim, fold_layernorm=fold, fold_min_rows=1)
    assert (unet.sts[0].fold is not None) == fold
    outs[fold] = unet(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(ctx))
    print(f"tiny U-Net, LayerNorm fold={fold}: rel {rel(outs[fold], ref):.3e}")
  assert rel(outs[True], ref) < 4e-2 and rel(outs[True], ref) <= rel(outs[False], ref) * 1.5


# ---- split-K reduce fused into the consuming GroupNorm (ldm_groupnorm_splitk) ---------------------
@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["f32", "bf16"])
@pytest.mark.parametrize("B,H,Cin,Cout,split", [(4, 8, 128, 320, 3), (2, 16, 64, 640, 2), (3, 4, 256, 1280, 4),
                                                (2, 8, 128, 960, 3),
                                                (2, 4, 128, 1280, 4)])      # HW = 16 < the rows a workgroup's threads span
def test_splitk_reduce_fused_into_groupnorm(dev, dtype, B, H, Cin, Cout, split):
  """conv3x3 (split-K, + bias + per-sample addend + residual) -> GroupNorm + SiLU.  The fused launch gives
  the SAME bits as reduce-then-GroupNorm (same summation order), for the stored product and for the
  normalised output; both agree with the oracle."""
  o = ops()
  x = rnd((B, H, H, Cin), 50).to(dtype)
  w_hwio = rnd((3, 3, Cin, Cout), 51, (9 * Cin) ** -0.5)
  bias, temb = rnd((Cout,), 52), rnd((B, Cout), 53)
  res = rnd((B, H, H, Cout), 54).to(dtype)
  gamma, beta = 1.0 + 0.3 * rnd((Cout,), 55), 0.2 * rnd((Cout,), 56)
  wt = L.conv_kernel(w_hwio.numpy(), dtype, dev)
  xd, resd = x.to(dev), res.to(dev)
  args = dict(bias=bias.to(dev), addend=temb.to(dev), residual=resd, split_k=split)
  # reference path: plain split-K conv (reduce launch), then GroupNorm
  y0 = torch.empty(B, H, H, Cout, dtype=dtype, device=dev)
  o.conv3x3(xd, wt, y0, **args)
  g0 = torch.empty_like(y0)
  o.groupnorm(y0, gamma.to(dev), beta.to(dev), g0, 1e-5, silu=True)
  # fused path
  y1 = torch.full((B, H, H, Cout), float("nan"), dtype=dtype, device=dev)
  pend = o.conv3x3(xd, wt, y1, defer_reduce=True, **args)
  assert isinstance(pend, o.PendingReduce) and not pend.done
  with pytest.raises(RuntimeError):                 # the workspace is busy until the product is completed
    o.conv3x3(xd, wt, torch.empty_like(y0), **args)
  g1 = torch.empty_like(y0)
  o.groupnorm(y1, gamma.to(dev), beta.to(dev), g1, 1e-5, silu=True, pending=pend)
  assert pend.done
  assert torch.equal(y1, y0) and torch.equal(g1, g0)
  # store_x=False: the product itself is not written
  y2 = torch.full((B, H, H, Cout), 7.0, dtype=dtype, device=dev)
  pend = o.conv3x3(xd, wt, y2, defer_reduce=True, **args)
  g2 = torch.empty_like(y0)
  o.groupnorm(y2, gamma.to(dev), beta.to(dev), g2, 1e-5, silu=True, pending=pend, store_x=False)
  assert torch.equal(g2, g0) and bool((y2 == 7.0).all())
  # finish(): the plain reduce of a deferred product
  y3 = torch.empty_like(y0)
  pend = o.conv3x3(xd, wt, y3, defer_reduce=True, **args)
  o.finish(pend)
  assert torch.equal(y3, y0)
  # oracle
  yr = O.conv2d(x.float(), w_hwio, bias) + temb[:, None, None, :] + res.float()
  gr = O.silu(O.group_norm(yr, gamma, beta, eps=1e-5))
  r = rel(g1, gr)
  print(f"split-K + GroupNorm fused [{dtype}] B={B} H={H} {Cin}->{Cout} split {split}: rel {r:.3e}")
  assert r < (2e-5 if dtype == torch.float32 else 1.5e-2)


def test_unet_deferred_reduces_equal_the_plain_path(dev):
  """The U-Net with split-K reduces fused into the consuming GroupNorms gives the same bits as with the
  separate reduce launches (forced split-K everywhere it can split, so the path is exercised on a tiny model)."""
  from ldm_tf2_amd.unet import UNet
  cfg = dict(model_channels=64, out_channels=4, num_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8)
  w = Wt.init_weights(Wt.unet_manifest(context_dim=128, **cfg), seed=2, mode="random", scope="unet")
  g = np.random.default_rng(3)
  x = g.standard_normal((4, 16, 16, 4)).astype(np.float32)
  ctx = g.standard_normal((4, 77, 128)).astype(np.float32)
  t = np.array([981, 981, 21, 500], dtype=np.int32)
  o = ops()
  outs = {}
  for dtype in (torch.float32, BF):
    for defer in (True, False):
      unet = UNet(**cfg, weights=w, dtype=dtype, device=dev, context_dim=128, defer_reduce=defer)
      n_fused = [0]
      orig = o.lib.ldm_groupnorm_splitk

      def counted(*a, _orig=orig, _n=n_fused):
        _n[0] += 1
        return _orig(*a)

      o.lib.ldm_groupnorm_splitk = counted
      try:
        outs[(dtype, defer)] = unet(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(ctx))
      finally:
        o.lib.ldm_groupnorm_splitk = orig
      print(f"tiny U-Net [{dtype}] defer_reduce={defer}: {n_fused[0]} fused reduce+GroupNorm launches")
      assert (n_fused[0] > 0) == defer
    assert torch.equal(outs[(dtype, True)], outs[(dtype, False)])
  ref = O.unet_forward(x, t, ctx, w)
  assert rel(outs[(torch.float32, True)], ref) < 5e-5


# ---- matrix-side softmax attention (ldm_attention_ms) -----------------------------------------------
def _ms_case(dev, R, H, Tq, Tk, spike=False, big_offset=0.0):
  o = ops()
  S, sp = 40, 48
  q = rnd((R, Tq, H, S), 60)
  k = rnd((R, Tk, H, S), 61)
  v = rnd((R, Tk, H, S), 62)
  if spike:
    k[0, Tk - 3, 0] = q[0, 5, 0] * 6.0               # one key that dominates query 5 in the LAST tile
  if big_offset:
    k = k + big_offset * q.mean(dim=1, keepdim=True) / 8.0
  scale = S ** -0.5
  qb, kb, vb = q.to(BF), k.to(BF), v.to(BF)
  logits = torch.einsum("nqhs,nchs->nhqc", qb.float(), kb.float()) * scale
  ref = torch.einsum("nhqc,nchs->nqhs", torch.softmax(logits, dim=3), vb.float())
  # what the projections deliver: q in the exp2 domain, 1.0 in dim 40 of k and in row 40 of V^T
  qd = torch.zeros(R, Tq, H, sp)
  qd[..., :S] = qb.float() * (scale * L.MS_LOG2E)
  kd = torch.zeros(R, Tk, H, sp)
  kd[..., :S] = kb.float()
  kd[..., L.MS_DIM] = 1.0
  tkp = (Tk + 7) // 8 * 8 + 8
  vt = torch.full((R, H * sp, tkp), float("nan"))
  vv = torch.zeros(R, Tk, H, sp)
  vv[..., :S] = vb.float()
  vv[..., L.MS_DIM] = 1.0
  vt[:, :, :Tk] = vv.reshape(R, Tk, H * sp).permute(0, 2, 1)
  out = torch.full((R, Tq, H * sp), float("nan"), dtype=BF, device=dev)
  o.attention(qd.reshape(R, Tq, H * sp).to(BF).to(dev), kd.reshape(R, Tk, H * sp).to(BF).to(dev),
              vt.to(BF).to(dev), out, H, sp, scale, matrix_softmax=True)
  got = out.reshape(R, Tq, H, sp).float().cpu()
  assert float(got[..., S:].abs().max()) == 0.0       # the padded output dims (incl. the denominator's) are zero
  # the plain kernel on the same bf16 q / k / v
  out0 = torch.empty(R, Tq, H * sp, dtype=BF, device=dev)
  q0 = torch.zeros(R, Tq, H, sp); q0[..., :S] = qb.float()
  k0 = torch.zeros(R, Tk, H, sp); k0[..., :S] = kb.float()
  vt0 = torch.zeros(R, H * sp, tkp)
  v0 = torch.zeros(R, Tk, H, sp); v0[..., :S] = vb.float()
  vt0[:, :, :Tk] = v0.reshape(R, Tk, H * sp).permute(0, 2, 1)
  o.attention(q0.reshape(R, Tq, H * sp).to(BF).to(dev), k0.reshape(R, Tk, H * sp).to(BF).to(dev), vt0.to(BF).to(dev),
              out0, H, sp, scale)
  r, r0 = rel(got[..., :S], ref), rel(out0.reshape(R, Tq, H, sp)[..., :S], ref)
  print(f"attention_ms R={R} H={H} Tq={Tq} Tk={Tk} spike={spike} offset={big_offset}: rel {r:.3e} (plain kernel {r0:.3e})")
  # q is rounded once more (after the exp2-domain scale): a little above the plain kernel's bf16 error
  assert r < 8e-3 and r < 2.5 * r0 + 1e-3


@pytest.mark.parametrize("R,H,Tq,Tk", [(2, 8, 256, 256), (2, 8, 192, 77), (1, 4, 1024, 1024), (1, 2, 100, 130),
                                       (1, 2, 1000, 333), (1, 2, 300, 64), (1, 2, 512, 77)])
def test_attention_matrix_softmax(dev, R, H, Tq, Tk):
  _ms_case(dev, R, H, Tq, Tk)


def test_attention_matrix_softmax_reference_moves(dev):
  """a dominating key in the last tile forces the reference to move late; a large common offset makes the
  first tile's maximum far from 0 in either direction"""
  _ms_case(dev, 1, 2, 256, 256, spike=True)
  _ms_case(dev, 1, 2, 128, 200, big_offset=40.0)
  _ms_case(dev, 1, 2, 128, 200, big_offset=-40.0)


# ---- fused feed-forward row-panel kernel (ldm_ffn_geglu) ----------------------------------------------
@pytest.mark.parametrize("M", [256, 1000, 4096])
def test_ffn_geglu_row_panel_kernel(dev, M):
  """x + Dense(a * gelu(g)) with (a | g) = Dense(LayerNorm(x)) (unet.py:313, :323-325, :335-338) as one launch,
  against the oracle and against the two-launch form (LayerNorm-folded GEGLU GEMM, then FF-out + residual)."""
  o = ops()
  C = 320
  x, gamma, beta = _ln_case(M, C, 70, 2.0)
  k1 = rnd((C, 8 * C), 71, C ** -0.5).numpy()
  b1 = rnd((8 * C,), 72).numpy()
  k2 = rnd((4 * C, C), 73, (4 * C) ** -0.5).numpy()
  b2 = rnd((C,), 74)
  y = O.dense(O.layer_norm(x.float(), gamma, beta, eps=1e-5), torch.from_numpy(k1), torch.from_numpy(b1))
  ref = x.float() + O.dense(y[:, :4 * C] * O.gelu(y[:, 4 * C:]), torch.from_numpy(k2), b2)
  gw, gb = L.geglu_kernel(k1, b1, torch.float32, "cpu")
  w1, cs, bb = L.ln_fold(gw, gamma.numpy(), beta.numpy(), gb.numpy(), BF, dev)
  w2 = L.dense_kernel(k2, BF, dev)
  aux = L.ffn_aux(cs, bb)
  xd = x.to(dev)
  out = torch.full((M, C), float("nan"), dtype=BF, device=dev)
  assert o.ffn_geglu_supported(xd)
  o.ffn_geglu(xd, w1, aux, w2, b2.to(dev), out, 1e-5)
  # two-launch form on the same folded weights
  ff = torch.empty(M, 4 * C, dtype=BF, device=dev)
  o.linear(xd, w1, ff, bias=bb, act=o.ACT_GEGLU, ln_fold=(cs, 1e-5))
  out2 = torch.empty(M, C, dtype=BF, device=dev)
  o.linear(ff, w2, out2, bias=b2.to(dev), residual=xd)
  r, r2, d = rel(out, ref), rel(out2, ref), rel(out, out2.float().cpu())
  print(f"ffn_geglu M={M}: rel {r:.3e} (two launches {r2:.3e}; fused vs two launches {d:.3e})")
  assert r < 4e-3 and r <= r2 * 1.2 + 1e-4 and d < 3e-3


@pytest.mark.parametrize("M", [128, 1000, 4096])
def test_st_tail_row_panel_kernel(dev, M):
  """o-projection + residual, feed-forward, proj_out + residual (unet.py:312-313, :363-365) as ONE launch, against
  the oracle and against the four launches it replaces."""
  o = ops()
  C, K0 = 320, 384
  att = rnd((M, K0), 80)
  r0, gamma, beta = _ln_case(M, C, 81, 2.0)
  r1 = rnd((M, C), 82)
  ko = rnd((K0, C), 83, K0 ** -0.5).numpy()
  bo = rnd((C,), 84)
  k1 = rnd((C, 8 * C), 85, C ** -0.5).numpy()
  b1 = rnd((8 * C,), 86).numpy()
  k2 = rnd((4 * C, C), 87, (4 * C) ** -0.5).numpy()
  b2 = rnd((C,), 88)
  kp = rnd((C, C), 89, C ** -0.5).numpy()
  bp = rnd((C,), 90)
  attb, r0b, r1b = att.to(BF), r0.to(BF), r1.to(BF)
  h = r0b.float() + O.dense(attb.float(), torch.from_numpy(ko), bo)
  y = O.dense(O.layer_norm(h, gamma, beta, eps=1e-5), torch.from_numpy(k1), torch.from_numpy(b1))
  y = h + O.dense(y[:, :4 * C] * O.gelu(y[:, 4 * C:]), torch.from_numpy(k2), b2)
  ref = r1b.float() + O.dense(y, torch.from_numpy(kp), bp)
  gw, gb = L.geglu_kernel(k1, b1, torch.float32, "cpu")
  w1, cs, bb = L.ln_fold(gw, gamma.numpy(), beta.numpy(), gb.numpy(), BF, dev)
  wo, w2, wp = (L.dense_kernel(k, BF, dev) for k in (ko, k2, kp))
  aux = L.ffn_aux(cs, bb)
  ad, r0d, r1d = attb.to(dev), r0b.to(dev), r1b.to(dev)
  out = torch.full((M, C), float("nan"), dtype=BF, device=dev)
  o.st_tail(ad, wo, bo.to(dev), r0d, w1, aux, w2, b2.to(dev), wp, bp.to(dev), r1d, out, 1e-5)
  hd = torch.empty(M, C, dtype=BF, device=dev)
  o.linear(ad, wo, hd, bias=bo.to(dev), residual=r0d)
  ff = torch.empty(M, 4 * C, dtype=BF, device=dev)
  o.linear(hd, w1, ff, bias=bb, act=o.ACT_GEGLU, ln_fold=(cs, 1e-5))
  yd = torch.empty(M, C, dtype=BF, device=dev)
  o.linear(ff, w2, yd, bias=b2.to(dev), residual=hd)
  out4 = torch.empty(M, C, dtype=BF, device=dev)
  o.linear(yd, wp, out4, bias=bp.to(dev), residual=r1d)
  r, r4, d = rel(out, ref), rel(out4, ref), rel(out, out4.float().cpu())
  print(f"st_tail M={M}: rel {r:.3e} (four launches {r4:.3e}; fused vs four launches {d:.3e})")
  assert torch.isfinite(out.float()).all()
  assert r < 5e-3 and r <= r4 * 1.2 + 1e-4 and d < 5e-3


@pytest.mark.parametrize("R,T,Tk,nan_pads", [(2, 128, 77, True), (3, 256, 77, False), (1, 384, 80, False), (2, 128, 5, True)])
def test_st_xtail_cross_attention_in_the_panel(dev, R, T, Tk, nan_pads):
  """Cross-attention (unet.py:273-291) + o-projection + feed-forward + proj_out as ONE launch: against the oracle
  and against ldm_attention_ms followed by ldm_st_tail on the same operands."""
  o = ops()
  C, H, S, sp = 320, 8, 40, 48
  K0, M = H * sp, R * T
  scale = S ** -0.5
  qb, kb, vb = (rnd(sh, sd).to(BF) for sh, sd in (((R, T, H, S), 100), ((R, Tk, H, S), 101), ((R, Tk, H, S), 102)))
  qb = qb * 2.0
  logits = torch.einsum("nqhs,nchs->nhqc", qb.float(), kb.float()) * scale
  att_ref = torch.einsum("nhqc,nchs->nqhs", torch.softmax(logits, dim=3), vb.float())
  qd = torch.zeros(R, T, H, sp); qd[..., :S] = qb.float() * (scale * L.MS_LOG2E)
  kd = torch.zeros(R, Tk, H, sp); kd[..., :S] = kb.float(); kd[..., L.MS_DIM] = 1.0
  ld = 80 if Tk > 8 else 88
  vt = torch.full((R, K0, ld), float("nan") if nan_pads else 0.0)
  vv = torch.zeros(R, Tk, H, sp); vv[..., :S] = vb.float(); vv[..., L.MS_DIM] = 1.0
  vt[:, :, :Tk] = vv.reshape(R, Tk, K0).permute(0, 2, 1)
  qd, kd, vt = qd.reshape(R, T, K0).to(BF).to(dev), kd.reshape(R, Tk, K0).to(BF).to(dev), vt.to(BF).to(dev)
  r0, gamma, beta = _ln_case(M, C, 103, 2.0)
  r0b, r1b = r0.to(BF), rnd((M, C), 104).to(BF)
  ko = torch.zeros(K0, C)                                    # rows of the padded dims are zero (layout.merge_kernel)
  ko.view(H, sp, C)[:, :S] = rnd((H, S, C), 105, (H * S) ** -0.5)
  bo, b2, bp = rnd((C,), 106), rnd((C,), 107), rnd((C,), 108)
  k1 = rnd((C, 8 * C), 109, C ** -0.5).numpy()
  b1 = rnd((8 * C,), 110).numpy()
  k2 = rnd((4 * C, C), 111, (4 * C) ** -0.5).numpy()
  kp = rnd((C, C), 112, C ** -0.5).numpy()
  att_p = torch.zeros(R, T, H, sp); att_p[..., :S] = att_ref
  h = r0b.float() + O.dense(att_p.reshape(M, K0), ko, bo)
  y = O.dense(O.layer_norm(h, gamma, beta, eps=1e-5), torch.from_numpy(k1), torch.from_numpy(b1))
  y = h + O.dense(y[:, :4 * C] * O.gelu(y[:, 4 * C:]), torch.from_numpy(k2), b2)
  ref = r1b.float() + O.dense(y, torch.from_numpy(kp), bp)
  gw, gb = L.geglu_kernel(k1, b1, torch.float32, "cpu")
  w1, cs, bb = L.ln_fold(gw, gamma.numpy(), beta.numpy(), gb.numpy(), BF, dev)
  wo, w2, wp = (L.dense_kernel(k, BF, dev) for k in (ko.numpy(), k2, kp))
  aux = L.ffn_aux(cs, bb)
  r0d, r1d = r0b.to(dev), r1b.to(dev)
  out = torch.full((M, C), float("nan"), dtype=BF, device=dev)
  o.st_xtail(qd, kd, vt, wo, bo.to(dev), r0d, w1, aux, w2, b2.to(dev), wp, bp.to(dev), r1d, out, 1e-5)
  att = torch.empty(R, T, K0, dtype=BF, device=dev)
  o.attention(qd, kd, vt, att, H, sp, scale, matrix_softmax=True)
  out2 = torch.empty(M, C, dtype=BF, device=dev)
  o.st_tail(att, wo, bo.to(dev), r0d, w1, aux, w2, b2.to(dev), wp, bp.to(dev), r1d, out2, 1e-5)
  r, r2, d = rel(out, ref), rel(out2, ref), rel(out, out2.float().cpu())
  print(f"st_xtail R={R} T={T} Tk={Tk}: rel {r:.3e} (attention + st_tail {r2:.3e}; fused vs those {d:.3e})")
  assert torch.isfinite(out.float()).all()
  assert r < 5e-3 and r <= r2 * 1.2 + 1e-4 and d < 5e-3


@pytest.mark.parametrize("R,T,Tk", [(2, 128, 77), (2, 384, 77), (1, 256, 9)])
def test_st_block_from_self_attention_output_to_block_output(dev, R, T, Tk):
  """o-projection + residual, LayerNorm-folded query projection, cross-attention, o-projection + residual,
  feed-forward, proj_out + residual (unet.py:310-313, :363-365) as ONE launch: against the oracle and against the
  launches it replaces (two GEMMs, then ldm_st_xtail)."""
  o = ops()
  C, H, S, sp = 320, 8, 40, 48
  K0, M = H * sp, R * T
  scale = S ** -0.5
  att1 = rnd((M, K0), 120).to(BF)
  r0, gamma, beta = _ln_case(M, C, 121, 2.0)
  r0b, r1b = r0.to(BF), rnd((M, C), 122).to(BF)
  g2, be2 = 1.0 + 0.3 * rnd((C,), 123), 0.2 * rnd((C,), 124)
  ko1 = rnd((K0, C), 125, K0 ** -0.5)
  bo1, bo2, b2, bp = (rnd((C,), sd) for sd in (126, 127, 128, 129))
  kq = rnd((C, H, S), 130, C ** -0.5)                        # the query projection has no bias (unet.py:262)
  kb, vb = rnd((R, Tk, H, S), 131).to(BF), rnd((R, Tk, H, S), 132).to(BF)
  ko2 = torch.zeros(K0, C)
  ko2.view(H, sp, C)[:, :S] = rnd((H, S, C), 133, (H * S) ** -0.5)
  k1 = rnd((C, 8 * C), 134, C ** -0.5).numpy()
  b1 = rnd((8 * C,), 135).numpy()
  k2 = rnd((4 * C, C), 136, (4 * C) ** -0.5).numpy()
  kp = rnd((C, C), 137, C ** -0.5).numpy()
  # oracle
  h1 = r0b.float() + O.dense(att1.float(), ko1, bo1)
  qr = torch.einsum("mc,chs->mhs", O.layer_norm(h1, g2, be2, eps=1e-5), kq).reshape(R, T, H, S)
  logits = torch.einsum("nqhs,nchs->nhqc", qr, kb.float()) * scale
  a2 = torch.einsum("nhqc,nchs->nqhs", torch.softmax(logits, dim=3), vb.float())
  a2p = torch.zeros(R, T, H, sp); a2p[..., :S] = a2
  h2 = h1 + O.dense(a2p.reshape(M, K0), ko2, bo2)
  y = O.dense(O.layer_norm(h2, gamma, beta, eps=1e-5), torch.from_numpy(k1), torch.from_numpy(b1))
  y = h2 + O.dense(y[:, :4 * C] * O.gelu(y[:, 4 * C:]), torch.from_numpy(k2), b2)
  ref = r1b.float() + O.dense(y, torch.from_numpy(kp), bp)
  # device operands
  wq_nk = torch.zeros(H, sp, C); wq_nk[:, :S] = kq.permute(1, 2, 0) * (scale * L.MS_LOG2E)
  wq, qcs, qb = L.ln_fold(wq_nk.reshape(K0, C), g2.numpy(), be2.numpy(), None, BF, dev)
  kd = torch.zeros(R, Tk, H, sp); kd[..., :S] = kb.float(); kd[..., L.MS_DIM] = 1.0
  vt = torch.full((R, K0, 80), float("nan"))
  vv = torch.zeros(R, Tk, H, sp); vv[..., :S] = vb.float(); vv[..., L.MS_DIM] = 1.0
  vt[:, :, :Tk] = vv.reshape(R, Tk, K0).permute(0, 2, 1)
  kd, vt = kd.reshape(R, Tk, K0).to(BF).to(dev), vt.to(BF).to(dev)
  gw, gb = L.geglu_kernel(k1, b1, torch.float32, "cpu")
  w1, cs, bb = L.ln_fold(gw, gamma.numpy(), beta.numpy(), gb.numpy(), BF, dev)
  wo1, wo2, w2, wp = (L.dense_kernel(k, BF, dev) for k in (ko1.numpy(), ko2.numpy(), k2, kp))
  aux = L.ffn_aux(cs, bb)
  ad, r0d, r1d = att1.reshape(R, T, K0).to(dev), r0b.to(dev), r1b.to(dev)
  bo1d, bo2d, b2d, bpd = (t_.to(dev) for t_ in (bo1, bo2, b2, bp))
  out = torch.full((M, C), float("nan"), dtype=BF, device=dev)
  o.st_block(ad, wo1, bo1d, r0d, wq, qcs, qb, kd, vt, wo2, bo2d, w1, aux, w2, b2d, wp, bpd, r1d, out, 1e-5)
  hd = torch.empty(M, C, dtype=BF, device=dev)
  o.linear(ad.view(M, K0), wo1, hd, bias=bo1d, residual=r0d)
  qd = torch.empty(R, T, K0, dtype=BF, device=dev)
  o.linear(hd, wq, qd.view(M, K0), bias=qb, ln_fold=(qcs, 1e-5))
  out2 = torch.empty(M, C, dtype=BF, device=dev)
  o.st_xtail(qd, kd, vt, wo2, bo2d, hd, w1, aux, w2, b2d, wp, bpd, r1d, out2, 1e-5)
  r, r2, d = rel(out, ref), rel(out2, ref), rel(out, out2.float().cpu())
  print(f"st_block R={R} T={T} Tk={Tk}: rel {r:.3e} (2 GEMMs + st_xtail {r2:.3e}; fused vs those {d:.3e})")
  assert torch.isfinite(out.float()).all()
  assert r < 6e-3 and r <= r2 * 1.2 + 1e-4 and d < 3e-3
  if R % 2 == 0:
    # in_rows (round 4): a classifier-free-guidance pair -- att / r0 / r1 hold only the first half of the rows, the
    # second half of the output reads the same input rows against ITS samples' context: the same bits as the launch
    # on explicitly duplicated inputs
    Rh, Mh = R // 2, M // 2
    dup = lambda t_: torch.cat([t_[:t_.shape[0] // 2], t_[:t_.shape[0] // 2]], 0).contiguous()
    out_d = torch.full((M, C), float("nan"), dtype=BF, device=dev)
    o.st_block(dup(ad), wo1, bo1d, dup(r0d), wq, qcs, qb, kd, vt, wo2, bo2d, w1, aux, w2, b2d, wp, bpd, dup(r1d), out_d, 1e-5)
    out_p = torch.full((M, C), float("nan"), dtype=BF, device=dev)
    o.st_block(ad[:Rh].contiguous(), wo1, bo1d, r0d[:Mh].contiguous(), wq, qcs, qb, kd, vt, wo2, bo2d, w1, aux, w2, b2d,
               wp, bpd, r1d[:Mh].contiguous(), out_p, 1e-5)
    assert torch.equal(out_p, out_d) and torch.equal(out_p[:Mh], out[:Mh]) and not torch.equal(out_p[Mh:], out[Mh:])


def test_st_block_at_the_benchmark_size_equals_the_per_layer_launches(dev):
  """BASELINE configs[2]'s 32x32 level (R = 32 rows of T = 1024 tokens): ldm_st_block against the eight per-layer
  launches it replaces (ldm_gemm x 7 with the LayerNorm folds, ldm_attention_ms) on the same operands; every
  sample / panel is covered, rows differ only by bf16 rounding of the intermediates."""
  o = ops()
  C, H, S, sp, R, T, Tk = 320, 8, 40, 48, 32, 1024, 77
  K0, M = H * sp, R * T
  g = torch.Generator().manual_seed(150)
  rn = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc)
  att1 = rn(R, T, K0).to(BF).to(dev)
  r0, r1 = rn(M, C).to(BF).to(dev), rn(M, C).to(BF).to(dev)
  wo1, wo2 = (rn(C, K0, sc=K0 ** -0.5).to(BF).to(dev) for _ in range(2))
  wp = rn(C, C, sc=C ** -0.5).to(BF).to(dev)
  bo1, bo2, b2, bp = (rn(C).to(dev) for _ in range(4))
  gam, bet = (1.0 + 0.3 * rn(C)).numpy(), (0.2 * rn(C)).numpy()
  wq_nk = torch.zeros(H, sp, C); wq_nk[:, :S] = rn(H, S, C, sc=C ** -0.5) * (S ** -0.5 * L.MS_LOG2E)
  wq, qcs, qb = L.ln_fold(wq_nk.reshape(K0, C), gam, bet, None, BF, dev)
  k1, b1 = rn(C, 8 * C, sc=C ** -0.5).numpy(), rn(8 * C).numpy()
  gw, gb = L.geglu_kernel(k1, b1, torch.float32, "cpu")
  w1, cs, bb = L.ln_fold(gw, gam, bet, gb.numpy(), BF, dev)
  aux = L.ffn_aux(cs, bb)
  w2 = rn(C, 4 * C, sc=(4 * C) ** -0.5).to(BF).to(dev)
  kd = torch.zeros(R, Tk, H, sp); kd[..., :S] = rn(R, Tk, H, S); kd[..., L.MS_DIM] = 1.0
  vv = torch.zeros(R, Tk, H, sp); vv[..., :S] = rn(R, Tk, H, S); vv[..., L.MS_DIM] = 1.0
  vt = torch.zeros(R, K0, 80); vt[:, :, :Tk] = vv.reshape(R, Tk, K0).permute(0, 2, 1)
  kd, vt = kd.reshape(R, Tk, K0).to(BF).to(dev), vt.to(BF).to(dev)
  out = torch.full((M, C), float("nan"), dtype=BF, device=dev)
  o.st_block(att1, wo1, bo1, r0, wq, qcs, qb, kd, vt, wo2, bo2, w1, aux, w2, b2, wp, bp, r1, out, 1e-5)
  h1, h2, y, ref = (torch.empty(M, C, dtype=BF, device=dev) for _ in range(4))
  q, a2 = torch.empty(R, T, K0, dtype=BF, device=dev), torch.empty(R, T, K0, dtype=BF, device=dev)
  ff = torch.empty(M, 4 * C, dtype=BF, device=dev)
  o.linear(att1.view(M, K0), wo1, h1, bias=bo1, residual=r0)
  o.linear(h1, wq, q.view(M, K0), bias=qb, ln_fold=(qcs, 1e-5))
  o.attention(q, kd, vt, a2, H, sp, S ** -0.5, matrix_softmax=True)
  o.linear(a2.view(M, K0), wo2, h2, bias=bo2, residual=h1)
  o.linear(h2, w1, ff, bias=bb, act=o.ACT_GEGLU, ln_fold=(cs, 1e-5))
  o.linear(ff, w2, y, bias=b2, residual=h2)
  o.linear(y, wp, ref, bias=bp, residual=r1)
  assert torch.isfinite(out.float()).all()
  d = (out.float() - ref.float()).view(R, T, C)
  per_sample = (d.norm(dim=(1, 2)) / ref.float().view(R, T, C).norm(dim=(1, 2))).cpu()
  print(f"st_block at M = {M}: rel per sample max {per_sample.max():.3e}, min {per_sample.min():.3e}")
  assert float(per_sample.max()) < 6e-3


def test_row_panel_launches_reject_what_they_cannot_do(dev):
  """Loud errors, nothing launched: other widths, f32, query rows per sample not a multiple of 128, too many keys, an
  output that aliases an input of ldm_st_block (out is its scratch)."""
  from ldm_tf2_amd._lib import LdmHipError
  o = ops()
  C, K0 = 320, 384
  z = lambda *sh, dt=BF: torch.zeros(*sh, dtype=dt, device=dev)
  w1, aux, w2, b = z(8 * C, C), z(8 * C * 2, dt=torch.float32), z(C, 4 * C), z(C, dt=torch.float32)
  wo, wp, wq, qv = z(C, K0), z(C, C), z(K0, C), z(K0, dt=torch.float32)
  x = z(256, C)
  with pytest.raises(LdmHipError):                                   # f32 rows
    o.ffn_geglu(z(256, C, dt=torch.float32), w1, aux, w2, b, z(256, C, dt=torch.float32), 1e-5)
  assert not o.ffn_geglu_supported(z(256, 640))                      # other widths: the per-layer launches
  q, ck, cv = z(2, 128, K0), z(2, 77, K0), z(2, K0, 80)
  out = z(256, C)
  o.st_xtail(q, ck, cv, wo, b, x, w1, aux, w2, b, wp, b, x, out, 1e-5)           # the accepted form
  with pytest.raises(LdmHipError):                                   # 96 query rows per sample
    o.st_xtail(z(2, 96, K0), ck, cv, wo, b, z(192, C), w1, aux, w2, b, wp, b, z(192, C), z(192, C), 1e-5)
  with pytest.raises(LdmHipError):                                   # 96 keys: more than one key tile
    o.st_xtail(q, z(2, 96, K0), z(2, K0, 96), wo, b, x, w1, aux, w2, b, wp, b, x, out, 1e-5)
  with pytest.raises(LdmHipError):                                   # out aliases the residual
    o.st_block(q, wo, b, x, wq, qv, qv, ck, cv, wo, b, w1, aux, w2, b, wp, b, x, x, 1e-5)
  torch.cuda.synchronize()


@pytest.mark.parametrize("fused,tail,block", [(True, True, True), (True, True, False), (True, False, False), (False, False, False)])
def test_unet_fused_ffn_matches_the_unfused_unet(dev, fused, tail, block):
  """A C = 320 U-Net level through ldm_st_block / ldm_st_xtail / ldm_ffn_geglu (ffn_min_rows=1) against the oracle
  and the unfused launches."""
  from ldm_tf2_amd.unet import UNet
  cfg = dict(model_channels=320, out_channels=4, num_blocks=1, channel_mult=(1,), num_heads=8)
  ctx_dim = 128
  w = Wt.init_weights(Wt.unet_manifest(context_dim=ctx_dim, **cfg), seed=7, mode="random", scope="unet")
  g = np.random.default_rng(8)
  x = g.standard_normal((2, 32, 32, 4)).astype(np.float32)
  ctx = g.standard_normal((2, 77, ctx_dim)).astype(np.float32)
  t = np.array([981, 21], dtype=np.int32)
  ref = O.unet_forward(x, t, ctx, w, num_heads=8)
  base = UNet(**cfg, weights=w, dtype=BF, device=dev, context_dim=ctx_dim, fold_min_rows=1, fused_ffn=False)
  out0 = base(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(ctx))
  unet = UNet(**cfg, weights=w, dtype=BF, device=dev, context_dim=ctx_dim, fold_min_rows=1, fused_ffn=fused,
              fused_tail=tail, fused_block=block, ffn_min_rows=1)
  assert unet.sts[0].ffn_aux is not None and unet.sts[0].ms
  calls = {"ldm_ffn_geglu": 0, "ldm_st_tail": 0, "ldm_st_xtail": 0, "ldm_st_block": 0}
  origs = {k: getattr(ops().lib, k) for k in calls}
  for k in calls:
    def counted(*a, _orig=origs[k], _k=k):
      calls[_k] += 1
      return _orig(*a)
    setattr(ops().lib, k, counted)
  try:
    out = unet(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(ctx))
  finally:
    for k in calls:
      setattr(ops().lib, k, origs[k])
  assert (calls["ldm_st_block"] > 0) == (fused and tail and block)
  assert (calls["ldm_st_xtail"] > 0) == (fused and tail and not block)
  assert (calls["ldm_ffn_geglu"] > 0) == (fused and not tail) and calls["ldm_st_tail"] == 0
  print(f"C=320 U-Net, fused feed-forward={fused} tail={tail} block={block}: rel {rel(out, ref):.3e} (unfused {rel(out0, ref):.3e}; {calls})")
  assert rel(out, ref) < 4e-2 and rel(out, ref) <= rel(out0, ref) * 1.5
