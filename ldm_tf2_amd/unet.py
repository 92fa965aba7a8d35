"""Conditioning U-Net on the HIP path -- host side.

Mirrors the reference's `UNet` operator (unet.py:51-138): same constructor
kwargs (= the YAML `unet` section), same call contract
`unet(x f32[R,h,w,4], t int32[R], context [R,77,D]) -> f32[R,h,w,4]`, NHWC.
The arithmetic runs in hand-written HIP kernels through the C ABI
(include/ldm_hip.h); this file only walks the block structure and hands out
buffers.  There is no CPU path.

What the host does differently from a line-by-line transcription (none of it
changes results beyond float rounding order):
  * the skip concatenation (unet.py:135) is free: every skip tensor and every
    block output is written straight into its channel slice of the buffer the
    consuming output block reads;
  * the timestep MLP (unet.py:126-127) and the 22 ResBlock temb projections
    (unet.py:386) depend only on t: they run once per call as four skinny-Dense
    launches; when all rows share one t (the DDIM loop, model_runners.py:449)
    they run for a single row;
  * cross-attention K and V of the text context (unet.py:274-275 with
    context != None) are step-invariant: `set_context` computes them once;
  * nearest-2x upsample (unet.py:44) is fused into the following conv's gather.
"""
from __future__ import annotations

import numpy as np
import torch

from . import layout as L
from . import ops
from .weights import init_weights, unet_manifest

GN_EPS_RES = 1e-5   # unet.py:374,377,115
GN_EPS_ST = 1e-6    # unet.py:354
LN_EPS = 1e-5       # unet.py:304-306


class _Res:
  """ResidualBlock weights (unet.py:368-380)."""

  def __init__(self, w, p, dtype, dev):
    g = lambda n: w[p + "/" + n]
    self.cin, self.cout = g("conv2d_1/kernel").shape[2], g("conv2d_1/kernel").shape[3]
    self.gn1 = (L.vec(g("group_norm_1/gamma"), dev), L.vec(g("group_norm_1/beta"), dev))
    self.conv1 = (L.conv_kernel(g("conv2d_1/kernel"), dtype, dev), L.vec(g("conv2d_1/bias"), dev))
    self.temb_k, self.temb_b = g("dense/kernel"), g("dense/bias")   # gathered by the U-Net
    self.gn2 = (L.vec(g("group_norm_2/gamma"), dev), L.vec(g("group_norm_2/beta"), dev))
    self.conv2 = (L.conv_kernel(g("conv2d_2/kernel"), dtype, dev), L.vec(g("conv2d_2/bias"), dev))
    self.shortcut = self.conv2_sc = None
    if (p + "/shortcut/kernel") in w:
      self.shortcut = (L.dense_kernel(g("shortcut/kernel"), dtype, dev), L.vec(g("shortcut/bias"), dev))
      # the shortcut as extra K columns of the second convolution (ldm_gemm a2): [Cout, 9 Cout + Cin], bias sum
      self.conv2_sc = (L.conv_shortcut_kernel(g("conv2d_2/kernel"), g("shortcut/kernel"), dtype, dev),
                       L.vec(np.asarray(g("conv2d_2/bias"), dtype=np.float32) + np.asarray(g("shortcut/bias"), dtype=np.float32), dev))
    self.temb_off = 0


class _ST:
  """SpatialTransformer weights (unet.py:341-354, :295-306, :248-265, :317-338)."""

  def __init__(self, w, p, heads, dtype, dev, fold_ln=False, matrix_softmax=True):
    g = lambda n: w[p + "/" + n]
    c = g("dense1/kernel").shape[0]
    self.c, self.heads = c, heads
    self.s = c // heads
    self.sp = L.padded_head(self.s)
    sp = self.sp
    self.gn = (L.vec(g("groupnorm/gamma"), dev), L.vec(g("groupnorm/beta"), dev))
    self.proj_in = (L.dense_kernel(g("dense1/kernel"), dtype, dev), L.vec(g("dense1/bias"), dev))
    self.proj_out = (L.dense_kernel(g("dense2/kernel"), dtype, dev), L.vec(g("dense2/bias"), dev))
    a1, a2 = p + "/block/att_layer1", p + "/block/att_layer2"
    self.qk1 = torch.cat([L.split_kernel(w[a1 + "/query/kernel"], sp, dtype, dev),
                          L.split_kernel(w[a1 + "/key/kernel"], sp, dtype, dev)], 0).contiguous()
    self.v1 = L.split_kernel(w[a1 + "/value/kernel"], sp, dtype, dev)
    self.qkv1 = torch.cat([self.qk1, self.v1], 0).contiguous()      # one launch: q | k | v (v stored as V^T)
    self.o1 = (L.merge_kernel(w[a1 + "/output/kernel"], sp, dtype, dev), L.vec(w[a1 + "/output/bias"], dev))
    self.q2 = L.split_kernel(w[a2 + "/query/kernel"], sp, dtype, dev)
    self.k2 = L.split_kernel(w[a2 + "/key/kernel"], sp, dtype, dev)
    self.v2 = L.split_kernel(w[a2 + "/value/kernel"], sp, dtype, dev)
    self.o2 = (L.merge_kernel(w[a2 + "/output/kernel"], sp, dtype, dev), L.vec(w[a2 + "/output/bias"], dev))
    self.geglu = L.geglu_kernel(g("block/ffn/geglu/kernel"), g("block/ffn/geglu/bias"), dtype, dev)
    self.ff_out = (L.dense_kernel(g("block/ffn/dense/kernel"), dtype, dev), L.vec(g("block/ffn/dense/bias"), dev))
    self.ln = [(L.vec(g(f"block/layernorm{i}/gamma"), dev), L.vec(g(f"block/layernorm{i}/beta"), dev))
               for i in (1, 2, 3)]
    # bf16: FF-out and proj_out as one product over (hidden | residual stream) (layout.ff_proj_fold; per-layer path)
    self.ffp = None
    if fold_ln:
      self.ffp = L.ff_proj_fold(g("block/ffn/dense/kernel"), g("block/ffn/dense/bias"), g("dense2/kernel"),
                                g("dense2/bias"), dtype, dev)
    # bf16: the three LayerNorms folded into the projections they feed (ldm_gemm ln_cs): (w', cs, b')
    self.fold = None
    # matrix-side softmax (ldm_attention_ms): 40-wide heads padded to 48, together with the fold (the
    # folded projections' biases carry the ones of the padded rows, their weights the exp2-domain scale)
    self.ms = bool(fold_ln and matrix_softmax and self.s == L.MS_DIM and sp == 48)
    self.ms_kbias = self.ms_vbias = None
    if fold_ln:
      lnp = [(g(f"block/layernorm{i}/gamma"), g(f"block/layernorm{i}/beta")) for i in (1, 2, 3)]
      f32, cpu = torch.float32, "cpu"
      hs = heads * sp
      qk_f = torch.cat([L.split_kernel(w[a1 + "/query/kernel"], sp, f32, cpu),
                        L.split_kernel(w[a1 + "/key/kernel"], sp, f32, cpu)], 0)
      gw, gb = L.geglu_kernel(g("block/ffn/geglu/kernel"), g("block/ffn/geglu/bias"), f32, cpu)
      qk_scale = qk_ones = v_ones = q_scale = None
      if self.ms:
        c2 = float(self.s) ** -0.5 * L.MS_LOG2E                  # unet.py:281 scale, in the exp2 domain
        qk_scale = torch.cat([torch.full((hs,), c2), torch.ones(hs)])
        qk_ones = L.ms_ones(heads, sp, offset=hs)                 # K[..., 40] = 1
        v_ones = L.ms_ones(heads, sp)                             # V^T row 40 = 1
        q_scale = torch.full((hs,), c2)
        self.ms_kbias, self.ms_vbias = v_ones.to(dev), v_ones.clone().to(dev)     # cross-attention K / V of the context
      v_f = L.split_kernel(w[a1 + "/value/kernel"], sp, f32, cpu)
      self.fold = dict(
          # q | k | v as ONE launch (row-major q | k, V^T transposed: ldm_gemm out2 with n_split = 2 heads Sp)
          qkv1=L.ln_fold(torch.cat([qk_f, v_f], 0), lnp[0][0], lnp[0][1], None, dtype, dev,
                         row_scale=None if qk_scale is None else torch.cat([qk_scale, torch.ones(hs)]),
                         bias_extra=None if qk_ones is None else torch.cat([qk_ones, v_ones])),
          qk1=L.ln_fold(qk_f, lnp[0][0], lnp[0][1], None, dtype, dev, row_scale=qk_scale, bias_extra=qk_ones),
          v1=L.ln_fold(L.split_kernel(w[a1 + "/value/kernel"], sp, f32, cpu), lnp[0][0], lnp[0][1], None, dtype, dev,
                       bias_extra=v_ones),
          q2=L.ln_fold(L.split_kernel(w[a2 + "/query/kernel"], sp, f32, cpu), lnp[1][0], lnp[1][1], None, dtype, dev,
                       row_scale=q_scale),
          geglu=L.ln_fold(gw, lnp[2][0], lnp[2][1], gb.numpy(), dtype, dev))
    # the whole feed-forward as one row-panel launch (ldm_ffn_geglu): C = 320 blocks, with the fold
    self.ffn_aux = None
    if self.fold is not None and c == 320:
      self.ffn_aux = L.ffn_aux(self.fold["geglu"][1], self.fold["geglu"][2])
    self.ctx_k = self.ctx_vt = None     # filled by UNet.set_context


class UNet:
  """Same kwargs as the reference constructor (unet.py:52-61).  Extra, build-only
  kwargs: `weights` (reference-layout float32 dict, weights.unet_manifest names;
  None = random init, the reference's behaviour when no checkpoint restores),
  `dtype` (torch.float32 | torch.bfloat16 storage), `device`, `context_dim`."""

  def __init__(self, model_channels=320, out_channels=4, num_blocks=2,
               attention_resolutions=(4, 2, 1), dropout_rate=0.1, channel_mult=(1, 2, 4, 4),
               num_heads=8, *, weights=None, dtype=torch.float32, device="cuda:0",
               context_dim=1280, init="keras", seed=2, fuse_layernorm=False, fuse_qkv=True,
               split_qkv=True, small_conv_out=False, fold_layernorm=True, fold_min_rows=2048,
               defer_reduce=True, matrix_softmax=True, gn_single_launch=True, fused_ffn=True, ffn_min_rows=24576, fused_tail=True, fused_xattn=True, fused_block=True, lanes=1, lane_levels=None, shared_prefix=True, merge_qkv=True, merge_qkv_max_rows=1 << 30,
               merge_shortcut=True, merge_ffproj=True, block_min_rows=12288):
    # fuse_layernorm: the transformer blocks' LayerNorms come out of the producing GEMM's epilogue
    # where its tile holds whole rows (C = 320).  Measured on MI355X at R=32: 11.02 vs 10.95 ms per
    # step -- the whole-row 128x320 tile (one workgroup per CU) plus the extra epilogue pass cost
    # slightly more than the 15 LayerNorm launches they replace.  Opt-in (parity-tested).
    self._fuse_ln = bool(fuse_layernorm)
    # fuse_qkv: the self-attention q|k and v projections as one GEMM launch with a transposed second
    # output (ldm_gemm out2)
    self._fuse_qkv = bool(fuse_qkv)
    # bf16: q|k and v as two persistent-kernel launches (v stored transposed); split_qkv=False: A/B
    self._split_qkv = dtype == torch.bfloat16 and bool(split_qkv)
    self._small_conv_out = bool(small_conv_out)
    # bf16: each LayerNorm of the transformer blocks (unet.py:309-313) is folded into the projection it
    # feeds -- LN(x) W^T = rstd (x (gamma (.) W)^T - mean colsum) + W beta -- and the persistent GEMM derives
    # the row statistics from the A tiles it stages anyway: 48 LayerNorm launches and their normalised
    # copies of the residual stream disappear.  Only where the launch has enough rows to fill the chip
    # on the persistent kernel (fold_min_rows); float32 keeps the separate LayerNorm (parity mode).
    self._fold_ln = dtype == torch.bfloat16 and bool(fold_layernorm) and not self._fuse_ln
    self._fold_min_rows = int(fold_min_rows)
    self._matrix_softmax = bool(matrix_softmax)   # ldm_attention_ms on the 40-wide heads (with the fold; A/B: False)
    self._fused_ffn = bool(fused_ffn)             # ldm_ffn_geglu on the C = 320 blocks (needs the fold; A/B: False)
    self._fused_tail = bool(fused_tail)           # ... with the o-projection before and proj_out after it (ldm_st_tail)
    self._fused_block = bool(fused_block)         # ... and o1-projection + query projection in front (ldm_st_block)
    self._fused_xattn = bool(fused_xattn)         # ... and the cross-attention in front of the tail (ldm_st_xtail)
    self._block_min_rows = int(block_min_rows)    # ... ldm_st_block alone from 192 panels of 64 rows on
    self._ffn_min_rows = int(ffn_min_rows)        # ... from 192 panels of 128 rows on (3/4 of the CUs busy)
    self._gn_single = bool(gn_single_launch)      # False: partial-sums + apply launches everywhere (A/B)
    self._defer_reduce = bool(defer_reduce)   # split-K reduces fused into the consuming GroupNorm (A/B: False)
    # lanes: the rows of one evaluation walked as `lanes` coarse, independent branches (no op of the U-Net
    # crosses rows, unet.py:118-138): branch i takes rows [i R / lanes, (i+1) R / lanes) on its own stream with
    # its own scratch and split-K workspace -- ONE fork after the timestep MLP, ONE join before the caller's
    # next launch -- so one branch's latency-bound launches run beside the other's convolutions.  Each
    # branch runs the launch plans of ITS row count.
    self._merge_ffproj = bool(merge_ffproj)       # FF-out + proj_out as one folded product on the per-layer path (bf16; A/B: False)
    self._merge_shortcut = bool(merge_shortcut)   # ResBlock shortcut inside its second convolution's K loop (A/B: False)
    self._merge_qkv = bool(merge_qkv)             # LayerNorm-folded q | k | V^T as one launch (A/B: False = two)
    # (a row limit from the first form of the launch, whose workgroups lay on one side of n_split and could not be
    # balanced at M = 32768: 6 + 3 n-tiles on 2 workgroups per panel; ranges may straddle n_split now)
    self._merge_qkv_max_rows = int(merge_qkv_max_rows)
    self._shared_prefix = bool(shared_prefix)     # forward(paired_rows=True): the CFG pair's common prefix once (A/B: False)
    self._lanes = max(1, int(lanes))
    # lane_levels = L: only the levels from L down (the downsample conv into level L .. the upsample conv out of it)
    # are branched, the full-resolution levels run unbranched on all rows; None: the whole evaluation
    self._lane_levels = None if lane_levels is None else int(lane_levels)
    self._lane_state = {}
    self._rows = None                     # a lane's slice of the rows (context K / V^T); None = all rows
    self._pend = None
    self._model_channels = model_channels
    self._out_channels = out_channels
    self._num_blocks = num_blocks
    self._attention_resolutions = attention_resolutions   # stored, never read (unet.py:66)
    self._dropout_rate = dropout_rate                     # inference: dropout inactive
    self._channel_mult = tuple(channel_mult)
    self._num_heads = num_heads
    self.dtype, self.device = dtype, torch.device(device)
    self.manifest = unet_manifest(model_channels, out_channels, num_blocks, self._channel_mult,
                                  num_heads, context_dim=context_dim)
    if weights is None:
      weights = init_weights(self.manifest, seed=seed, mode=init, scope="unet")
    missing = [k for k in self.manifest if k not in weights]
    if missing:
      raise KeyError(f"UNet weights missing {len(missing)} tensors, e.g. {missing[:3]}")
    self.buf = L.Buffers(self.device)
    self._ws = ops.new_workspace(self.device)   # this model's split-K workspace (ops.workspace_scope)
    self._compile(weights)

  # ---- build ---------------------------------------------------------------------
  def _compile(self, w):
    dt, dev, H = self.dtype, self.device, self._num_heads
    mc = self._model_channels
    self.conv_in = (L.vec(w["conv_in/kernel"], dev), L.vec(w["conv_in/bias"], dev))
    self.time1 = (L.dense_kernel(w["time_dense1/kernel"], dt, dev), L.vec(w["time_dense1/bias"], dev))
    self.time2 = (L.dense_kernel(w["time_dense2/kernel"], dt, dev), L.vec(w["time_dense2/bias"], dev))
    res_all = []

    def mk_res(p):
      r = _Res(w, p, dt, dev)
      res_all.append(r)
      return r

    def mk_st(p):
      return (_ST(w, p, H, dt, dev, fold_ln=self._fold_ln, matrix_softmax=self._matrix_softmax)
              if (p + "/dense1/kernel") in w else None)

    self.in_blocks, self.skip_ch, self.skip_lvl = [], [mc], [0]
    lvl, i = 0, 0
    while any(k.startswith(f"input_blocks/{i}/") for k in w):
      p = f"input_blocks/{i}"
      if (p + "/downsample/conv/kernel") in w:
        k = w[p + "/downsample/conv/kernel"]
        self.in_blocks.append(("down", L.conv_kernel(k, dt, dev), L.vec(w[p + "/downsample/conv/bias"], dev)))
        lvl += 1
        self.skip_ch.append(k.shape[3])
      else:
        r = mk_res(p + "/residual")
        self.in_blocks.append(("res", r, mk_st(p + "/spatial_transformer")))
        self.skip_ch.append(r.cout)
      self.skip_lvl.append(lvl)
      i += 1
    self.mid = (mk_res("middle_block/residual1"), mk_st("middle_block/spatial_transformer"),
                mk_res("middle_block/residual2"))
    self.out_blocks = []
    i = 0
    while any(k.startswith(f"output_blocks/{i}/") for k in w):
      p = f"output_blocks/{i}"
      up = None
      if (p + "/upsample/conv/kernel") in w:
        up = (L.conv_kernel(w[p + "/upsample/conv/kernel"], dt, dev), L.vec(w[p + "/upsample/conv/bias"], dev))
      self.out_blocks.append((mk_res(p + "/residual"), mk_st(p + "/spatial_transformer"), up))
      i += 1
    assert len(self.out_blocks) == len(self.skip_ch), "U-Net skip structure mismatch"
    self.gn_out = (L.vec(w["groupnorm/gamma"], dev), L.vec(w["groupnorm/beta"], dev))
    self.conv_out = (L.vec(w["conv_out/kernel"], dev), L.vec(w["conv_out/bias"], dev))
    # bf16: the 320 -> 4 output conv as an implicit-GEMM launch too (N = 4 of a 64-column tile is
    # wasted MFMA work, but 3 GFLOP on the matrix cores beat the 8-lanes-per-pixel FMA kernel 4x);
    # small_conv_out=True: the scalar kernel (A/B)
    self.conv_out_mm = (L.conv_kernel(w["conv_out/kernel"], dt, dev)
                        if dt == torch.bfloat16 and not self._small_conv_out else None)
    # all ResBlock temb projections as ONE skinny Dense [sum(Cout), 4*mc]
    off, ks, bs = 0, [], []
    for r in res_all:
      r.temb_off = off
      off += r.cout
      ks.append(L.dense_kernel(r.temb_k, dt, dev))
      bs.append(L.vec(r.temb_b, dev))
      r.temb_k = r.temb_b = None
    self.temb_all = (torch.cat(ks, 0).contiguous(), torch.cat(bs, 0).contiguous())
    self.temb_total = off
    self.sts = [b[2] for b in self.in_blocks if b[0] == "res" and b[2] is not None]
    self.sts += [self.mid[1]] + [b[1] for b in self.out_blocks if b[1] is not None]

  # ---- step-invariant cross-attention K / V ------------------------------------------
  def set_context(self, context):
    """Projects the text context through every cross-attention key/value layer once
    (unet.py:274-275).  The projections land in buffers owned by this model (one set per
    context shape), so a captured step graph stays valid across calls; callers invoke this
    EVERY time they are handed a context (32 small GEMMs) -- nothing is cached on tensor
    identity, because a freed context's address is routinely reused by the next one."""
    if context.dtype != self.dtype:
      c2 = self.buf.get("ctx_cast", context.shape, self.dtype)
      ops.cast(context, c2)
      context = c2
    R, Tk, _ = context.shape
    tkp = (Tk + 7) // 8 * 8
    with ops.workspace_scope(self._ws):
      for n, st in enumerate(self.sts):
        hs = st.heads * st.sp
        st.ctx_k = self.buf.get(f"ctxk{n}", (R, Tk, hs), self.dtype)
        st.ctx_vt = self.buf.get(f"ctxv{n}", (R, hs, tkp), self.dtype, zero=True)
        # (matrix-side softmax blocks: 1.0 in the padded dim 40 of every head of K and V^T; inert for
        # the plain attention kernel, whose Q is zero there)
        ops.linear(context, st.k2, st.ctx_k, bias=st.ms_kbias)
        ops.bmm_nt(context, st.v2, st.ctx_vt, transposed_out=True, bias=st.ms_vbias)
    self._ctx_rows = R

  # ---- blocks ------------------------------------------------------------------------------
  # ---- split-K products whose reduce is fused into the GroupNorm that consumes them ---------------
  # A split-K convolution leaves float32 slabs; its reduce + epilogue launch is followed, almost
  # everywhere in the U-Net, by the GroupNorm of exactly that tensor.  `_conv_deferred` launches only the
  # main kernel and parks the slabs (self._pend); `_gn` completes them inside the GroupNorm launch
  # (ldm_groupnorm_splitk: same value, bit for bit, one launch and one bf16 round trip less).  Whatever
  # else is about to read the tensor or use the workspace calls `_flush` first (plain reduce).
  def _flush(self):
    if self._pend is not None:
      ops.finish(self._pend)
      self._pend = None

  def _conv_deferred(self, x, wt, out, **kw):
    self._flush()
    r = ops.conv3x3(x, wt, out, defer_reduce=self._defer_reduce, **kw)
    self._pend = r if isinstance(r, ops.PendingReduce) else None
    return out

  def _gn(self, x, gn, eps, silu, out, store_x=True):
    pend, self._pend = self._pend, None
    ops.groupnorm(x, gn[0], gn[1], out, eps, silu=silu, partial=self._gnp, pending=pend, store_x=store_x,
                  fused=None if self._gn_single else False)
    return out

  def _res(self, r, x, tall, out):
    B_, dt = self.buf, self.dtype
    R, h, w, _ = x.shape
    # GN1 first: it completes a deferred product that produced x (x is materialised by that launch)
    t0 = B_.get("gn", tuple(x.shape), dt)
    self._gn(x, r.gn1, GN_EPS_RES, True, t0)
    res = x
    merged = r.shortcut is not None and self._merge_shortcut
    if r.shortcut is not None and not merged:   # before conv1: the slabs of conv1 must survive until GN2
      res = B_.get("sc", (R, h, w, r.cout), dt)
      ops.linear(x, r.shortcut[0], res, bias=r.shortcut[1])
    h1 = B_.get("h1", (R, h, w, r.cout), dt)
    self._conv_deferred(t0, r.conv1[0], h1, bias=r.conv1[1], addend=tall[:, r.temb_off:r.temb_off + r.cout])
    t1 = B_.get("gn", (R, h, w, r.cout), dt)
    self._gn(h1, r.gn2, GN_EPS_RES, True, t1, store_x=False)      # h1 has no other reader (unet.py:388-390)
    # conv2's reduce is left to the next GroupNorm (the following block's first op) when that reads `out`
    if merged:
      # unet.py:393-397: shortcut(x) + conv2(...) as ONE product -- the shortcut's channels are extra K columns of
      # the convolution (ldm_gemm a2): no launch and no [M, Cout] tensor of its own
      self._conv_deferred(t1, r.conv2_sc[0], out, bias=r.conv2_sc[1], x2=x)
    else:
      self._conv_deferred(t1, r.conv2[0], out, bias=r.conv2[1], residual=res)
    return out

  def _pair_buf(self, tag, shape, dt, pair):
    """(view, full): scratch for a tensor of `shape`; for a CFG pair the buffer has twice the rows and `view` is
    its first half, so that leaving the common prefix is ONE copy of the first half onto the second (_dup_rows)."""
    if not pair:
      t = self.buf.get(tag, shape, dt)
      return t, t
    full = self.buf.get(tag + "_pair", (2 * shape[0],) + tuple(shape[1:]), dt)
    return full[:shape[0]], full

  def _dup_rows(self, full):
    """full[n:] = full[:n] (a CFG pair leaving its common prefix on a path without ldm_st_block's in_rows)."""
    n = full.shape[0] // 2
    ops.cast(full[:n], full[n:])
    return full

  def _st(self, st, x, out, pair=False, x_full=None):
    """`pair`: x holds the first half of out's rows and the second half is identical up to the first
    cross-attention (forward(paired_rows=True)); out and the context cover all rows."""
    B_, dt = self.buf, self.dtype
    R, h, w, c = x.shape
    T, hs = h * w, st.heads * st.sp
    scale = st.s ** -0.5
    t0 = B_.get("gn", (R, h, w, c), dt)
    self._gn(x, st.gn, GN_EPS_ST, False, t0)
    ha, ha_full = self._pair_buf("st_a", (R, T, c), dt, pair)
    ln = B_.get("st_ln", (R, T, c), dt)
    # each LayerNorm of the block normalises a row the preceding projection has just produced:
    # where the GEMM tile holds whole rows (C = 320) it is emitted by that GEMM's epilogue
    fuse_ln = self._fuse_ln and ops.linear_ln_supported(c, dt)
    lnp = lambda i: (st.ln[i][0], st.ln[i][1], ln, LN_EPS) if fuse_ln else None
    ops.linear(t0, st.proj_in[0], ha, bias=st.proj_in[1], ln=lnp(0))
    qk = B_.get("st_qk", (R, T, 2 * hs), dt)
    tp = (T + 7) // 8 * 8
    vt = B_.get("st_vt", (R, hs, tp), dt, zero=True)
    # LayerNorm folded into its consumer (bf16, launches with enough rows for the persistent kernel)
    fold = st.fold if (st.fold is not None and R * T >= self._fold_min_rows and T % 32 == 0) else None
    # self-attention (unet.py:309-310)
    if fold is not None and self._merge_qkv and R * T <= self._merge_qkv_max_rows and hs % 128 == 0:
      # q | k | V^T: one pass of the persistent kernel over the LayerNorm'ed rows (gemm3_kernel EPI bit 7)
      ops.linear(ha, fold["qkv1"][0], qk, bias=fold["qkv1"][2], ln_fold=(fold["qkv1"][1], LN_EPS), out2=vt)
    elif fold is not None:
      ops.linear(ha, fold["qk1"][0], qk, bias=fold["qk1"][2], ln_fold=(fold["qk1"][1], LN_EPS))
      ops.linear_t(ha, fold["v1"][0], vt, bias=fold["v1"][2], ln_fold=(fold["v1"][1], LN_EPS))
    elif not fuse_ln:
      ops.layernorm(ha, st.ln[0][0], st.ln[0][1], ln, LN_EPS)
    if fold is not None:
      pass
    elif self._split_qkv and ops.linear_t_supported(ln, st.v1, vt):
      # bf16: q|k row-major and v TRANSPOSED (straight into the attention kernel's V^T layout) as two
      # launches that can both take the persistent kernel (ops.linear_t)
      ops.linear(ln, st.qk1, qk)
      ops.linear_t(ln, st.v1, vt)
    elif self._fuse_qkv and T % 4 == 0:
      # q | k | v in ONE launch: q|k row-major, v straight into the attention kernel's V^T layout
      ops.linear(ln, st.qkv1, qk, out2=vt)
    else:
      ops.linear(ln, st.qk1, qk)
      ops.bmm_nt(ln, st.v1, vt, transposed_out=True)
    att, att_full = self._pair_buf("st_att", (R, T, hs), dt, pair)
    ms = fold is not None and st.ms
    ops.attention(qk[..., :hs], qk[..., hs:], vt, att, st.heads, st.sp, scale, matrix_softmax=ms)
    ctx_k, ctx_vt = (st.ctx_k, st.ctx_vt) if self._rows is None else (st.ctx_k[self._rows], st.ctx_vt[self._rows])
    Ro = ctx_k.shape[0]                   # rows of `out` (= 2 R for a pair)
    assert Ro == (2 * R if pair else R) and out.shape[0] == Ro
    panel = (fold is not None and st.ffn_aux is not None and self._fused_ffn and Ro * T >= self._ffn_min_rows)
    xtail = (panel and self._fused_tail and self._fused_xattn and ms and hs == 384 and T % 128 == 0
             and ctx_k.shape[1] <= 80 and ctx_vt.shape[2] >= 80)
    # ldm_st_block has a 64-row panel form for launches below 192 panels of 128 rows (round 4): it pays from 192
    # panels of 64 rows on (`block_min_rows`), where the other row-panel launches (128-row panels only) do not
    blk = (self._fused_block and fold is not None and st.ffn_aux is not None and self._fused_ffn and self._fused_tail
           and self._fused_xattn and ms and hs == 384 and T % 128 == 0 and ctx_k.shape[1] <= 80 and ctx_vt.shape[2] >= 80
           and Ro * T >= min(self._ffn_min_rows, self._block_min_rows))
    if pair and not blk:
      # per-layer path: the pair leaves its common prefix here -- both halves get their copy of the self-attention's
      # output, the residual stream and the block input, and everything below covers all rows
      att, ha, x = self._dup_rows(att_full), self._dup_rows(ha_full), self._dup_rows(x_full)
      R = Ro
      ln = B_.get("st_ln", (R, T, c), dt)
      lnp = lambda i: (st.ln[i][0], st.ln[i][1], ln, LN_EPS) if fuse_ln else None
    hb = B_.get("st_b", (R, T, c), dt)
    q = B_.get("st_q", (R, T, hs), dt)
    if blk:
      # everything from the self-attention's output to the block's output: ONE row-panel launch (ldm_st_block)
      ops.st_block(att, st.o1[0], st.o1[1], ha, fold["q2"][0], fold["q2"][1], fold["q2"][2], ctx_k, ctx_vt,
                   st.o2[0], st.o2[1], fold["geglu"][0], st.ffn_aux, st.ff_out[0], st.ff_out[1], st.proj_out[0],
                   st.proj_out[1], x, out, LN_EPS)      # (pair: att / ha / x hold half of out's rows -> in_rows)
      return out
    ops.linear(att, st.o1[0], hb, bias=st.o1[1], residual=ha, ln=lnp(1))
    # cross-attention (unet.py:311-312)
    if fold is not None:
      ops.linear(hb, fold["q2"][0], q, bias=fold["q2"][2], ln_fold=(fold["q2"][1], LN_EPS))
    else:
      if not fuse_ln:
        ops.layernorm(hb, st.ln[1][0], st.ln[1][1], ln, LN_EPS)
      ops.linear(ln, st.q2, q)
    if xtail:
      # ... with the cross-attention itself in front of it, in place in the LDS panel
      ops.st_xtail(q, ctx_k, ctx_vt, st.o2[0], st.o2[1], hb, fold["geglu"][0], st.ffn_aux, st.ff_out[0],
                   st.ff_out[1], st.proj_out[0], st.proj_out[1], x, out, LN_EPS)
      return out
    ops.attention(q, ctx_k, ctx_vt, att, st.heads, st.sp, scale, matrix_softmax=ms)
    if panel and self._fused_tail and hs == 384:
      # o-projection + residual, LayerNorm -> GEGLU -> FF-out + residual, proj_out + residual: ONE row-panel
      # launch; the two intermediate residual-stream tensors live only in LDS (unet.py:312-313, :363-365)
      ops.st_tail(att, st.o2[0], st.o2[1], hb, fold["geglu"][0], st.ffn_aux, st.ff_out[0], st.ff_out[1],
                  st.proj_out[0], st.proj_out[1], x, out, LN_EPS)
      return out
    ops.linear(att, st.o2[0], ha, bias=st.o2[1], residual=hb, ln=lnp(2))
    # GEGLU feed-forward (unet.py:313, :323-325, :335-338)
    if panel:
      # LayerNorm -> GEGLU -> FF-out + residual as ONE row-panel launch: the [R*T, 4C] hidden activation
      # never reaches HBM (unet.py:313)
      ops.ffn_geglu(ha, fold["geglu"][0], st.ffn_aux, st.ff_out[0], st.ff_out[1], hb, LN_EPS)
      ops.linear(hb, st.proj_out[0], out, bias=st.proj_out[1], residual=x)
      return out
    ff = B_.get("st_ff", (R, T, 4 * c), dt)
    if fold is not None:
      ops.linear(ha, fold["geglu"][0], ff, bias=fold["geglu"][2], act=ops.ACT_GEGLU,
                 ln_fold=(fold["geglu"][1], LN_EPS))
    else:
      if not fuse_ln:
        ops.layernorm(ha, st.ln[2][0], st.ln[2][1], ln, LN_EPS)
      ops.linear(ln, st.geglu[0], ff, bias=st.geglu[1], act=ops.ACT_GEGLU)
    if st.ffp is not None and self._merge_ffproj:
      # y = h + FF-out(ff) and proj_out(y) + x are two linear layers with a residual between them: ONE product over
      # (ff | h) with the folded weights (Wp W2 | Wp) -- no proj_out launch, no y tensor (unet.py:313, :363-365)
      # (its split-K reduce is left to the GroupNorm that reads `out` next -- the following ResBlock's, on the way
      # down -- like a convolution's; whatever else comes next flushes it as a plain reduce)
      self._flush()
      r = ops.linear(ff, st.ffp[0], out, bias=st.ffp[1], residual=x, x2=ha, defer_reduce=self._defer_reduce)
      self._pend = r if isinstance(r, ops.PendingReduce) else None
      return out
    ops.linear(ff, st.ff_out[0], hb, bias=st.ff_out[1], residual=ha)
    ops.linear(hb, st.proj_out[0], out, bias=st.proj_out[1], residual=x)
    return out

  # ---- forward ---------------------------------------------------------------------------------
  def forward(self, x, t_rows=None, steps=None, index=None, out=None, shared_t=False, paired_rows=False,
              temb_table=None, pre_decrement=False):
    """x f32 [R,h,w,4].  Timestep either per row (`t_rows` int32 [R]) or, for the
    graph-replayed DDIM loop, `steps[*index]` for every row.  `shared_t=True` with
    t_rows declares that all rows carry t_rows[0].
    `paired_rows=True` declares that rows r and r + R/2 carry the SAME x and t and differ only in their context
    (the classifier-free-guidance batch concat([xt, xt]) of model_runners.py:449-452).  Nothing of the U-Net mixes
    rows and the context first enters at the first cross-attention (unet.py:311), so the launches in front of it
    -- the first ResBlock and the first transformer block up to its self-attention -- are the same numbers for both
    halves: they run once, on R/2 rows (`shared_prefix`, bf16 and float32).
    `temb_table` (from `temb_table(steps)`) with `index`: the step's temb projections are row *index of that table
    (one launch instead of four); `pre_decrement`: that launch first decrements *index, the DDIM loop's counter."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    R, h, w, _ = x.shape
    nlev = max(self.skip_lvl)
    assert h % (1 << nlev) == 0 and w % (1 << nlev) == 0, "latent size must divide by 2**levels"
    assert self.sts[0].ctx_k is not None and self._ctx_rows == R, "call set_context(context) first"
    if out is None:
      out = torch.empty(R, h, w, self._out_channels, dtype=torch.float32, device=self.device)
    if temb_table is not None:
      assert index is not None and temb_table.shape[1] == self.temb_total
      tall = self.buf.get("temb_all", (1, self.temb_total), torch.float32)
      ops.select_row(temb_table, index, tall, pre_decrement=pre_decrement)
    else:
      assert not pre_decrement
      tall = self._temb(R, t_rows, steps, index, shared_t)
    env = self._env(x, tall, out)
    env["pair"] = bool(paired_rows and self._shared_prefix and R % 2 == 0 and self._lanes == 1
                       and self.in_blocks[0][0] == "res" and self.in_blocks[0][2] is not None)
    prog = env["prog"]
    n = self._lanes if (self._lanes > 1 and R % self._lanes == 0) else 1
    if n == 1:
      self._segment(env, 0, len(prog), None)
      return out
    # coarse row branches over the step range [a, b): lane 0 on the caller's stream, the others on side streams of
    # their own (ONE fork, ONE join); under graph capture the side streams join the capture through the event
    # waits.  Steps outside the range run unbranched on all rows.
    a, b = self._lane_range(env)
    Rl = R // n
    cur = torch.cuda.current_stream(self.device)
    lanes = self._lane_views(n)
    if a > 0:
      self._segment(env, 0, a, None)
    for i in range(1, n):
      ln, side = lanes[i]
      side.wait_stream(cur)
      with torch.cuda.stream(side):
        ln._segment(env, a, b, slice(i * Rl, (i + 1) * Rl))
    lanes[0][0]._segment(env, a, b, slice(0, Rl))
    for i in range(1, n):
      cur.wait_stream(lanes[i][1])
    if b < len(prog):
      self._segment(env, b, len(prog), None)
    return out

  def _lane_range(self, env):
    """Step range the row branches cover: the whole evaluation (`lane_levels` None) or the levels from
    `lane_levels` down -- first step = the downsample conv into that level, last = the upsample conv out of it."""
    prog = env["prog"]
    if self._lane_levels is None:
      return 0, len(prog)
    lv = [st[-1] for st in prog]
    inside = [i for i, l in enumerate(lv) if l >= self._lane_levels]
    return (inside[0], inside[-1] + 1) if inside else (0, 0)

  def _lane_views(self, n):
    """Shallow views of this model, one per branch: the weights are shared, scratch buffers, split-K workspace and
    the deferred-product slot are the view's own (built once per lane count, before any capture needs them)."""
    views = self._lane_state.get(n)
    if views is None:
      import copy
      views = []
      for i in range(n):
        v = copy.copy(self)
        v.buf = L.Buffers(self.device)
        v._ws = ops.new_workspace(self.device)
        v._pend = None
        v._lanes, v._lane_state = 1, {}
        views.append((v, torch.cuda.Stream(self.device) if i else None))
      self._lane_state[n] = views
    return views

  def _segment(self, env, a, b, rows):
    """Steps [a, b) of the evaluation on the rows `rows` (None = all), with this view's scratch and workspace and
    the launch plans measured for this row count (ops.plan_scope)."""
    x = env["x"]
    nr = x.shape[0] if rows is None else rows.stop - rows.start
    self._rows = rows
    with ops.plan_scope(nr, x.shape[1], self.dtype), ops.workspace_scope(self._ws):
      self._pend = None
      self._gnp = self.buf.get("gn_partial", (nr * 128 * 32 * 2,), torch.float32)
      try:
        for st in env["prog"][a:b]:
          self._exec(env, st, rows)
      finally:
        self._flush()

  def temb_table(self, steps):
    """[len(steps), sum of Cout] float32: the temb projections of every timestep in `steps` (int32, device) -- the
    same launches, row by row the same arithmetic, as one evaluation makes for its own t (unet.py:125-127, :386).
    The DDIM loop builds it once and hands it to every step (forward(temb_table=...))."""
    n = steps.numel()
    tall = self._temb(n, steps.to(torch.int32).contiguous(), None, None, False, tag="tbl")
    return tall

  def _temb(self, R, t_rows, steps, index, shared_t, tag=""):
    """timestep embedding + MLP + all temb projections (unet.py:125-127, :386): [1 or R, sum of Cout] f32."""
    B_, mc = self.buf, self._model_channels
    f32 = torch.float32
    rt = 1 if (index is not None or shared_t) else R
    emb = B_.get("temb_sin" + tag, (rt, mc), f32)
    if index is not None:
      ops.time_embedding(emb, mc, steps=steps, index=index)
    else:
      ops.time_embedding(emb, mc, t_rows=t_rows)
    th = B_.get("temb_h" + tag, (rt, 4 * mc), f32)
    ops.gemv(emb, self.time1[0], self.time1[1], th, act_out=ops.ACT_SILU)
    temb = B_.get("temb" + tag, (rt, 4 * mc), f32)
    ops.gemv(th, self.time2[0], self.time2[1], temb)
    tall = B_.get("temb_all" + tag, (rt, self.temb_total), f32)
    ops.gemv(temb, self.temb_all[0], self.temb_all[1], tall, act_in=ops.ACT_SILU)
    return tall

  def _env(self, x, tall, out):
    """The evaluation as a flat program over buffers every branch shares: the skip / concat buffers `cats`
    (cat[j] = input of output block j = [previous output | skip n_in - j]), the final feature map, x, out.
    A step is (kind, index, level); a branch executes a range of steps on its row slice of these buffers."""
    R, h, w, _ = x.shape
    B_, dt = self.buf, self.dtype
    n_in = len(self.in_blocks)
    prev_ch = [self.mid[2].cout] + [b[0].cout for b in self.out_blocks[:-1]]
    cats = []
    for j in range(len(self.out_blocks)):
      i = n_in - j
      lv = self.skip_lvl[i]
      cats.append(B_.get(f"cat{j}", (R, h >> lv, w >> lv, prev_ch[j] + self.skip_ch[i]), dt))
    final = B_.get("final", (R, h, w, self.out_blocks[-1][0].cout), dt)
    prog = [("conv_in", 0, 0)]
    for i in range(n_in):
      prog.append(("in", i, self.skip_lvl[i + 1]))
    top = max(self.skip_lvl)
    prog.append(("mid", 0, top))
    for j in range(len(self.out_blocks)):
      prog.append(("out", j, self.skip_lvl[n_in - j]))
    prog.append(("final", 0, 0))
    return dict(x=x, tall=tall, out=out, cats=cats, final=final, prev_ch=prev_ch, prog=prog)

  def _exec(self, env, step, rows):
    kind, idx, _ = step
    B_, dt = self.buf, self.dtype
    sl = (lambda t: t) if rows is None else (lambda t: t[rows])
    x, out, cats, prev_ch = sl(env["x"]), sl(env["out"]), env["cats"], env["prev_ch"]
    tall = env["tall"] if env["tall"].shape[0] == 1 else sl(env["tall"])
    R = x.shape[0]
    n_in = len(self.in_blocks)
    skip_dst = lambda i: sl(cats[n_in - i])[..., prev_ch[n_in - i]:]
    if kind == "conv_in":
      if env.get("pair") and rows is None:
        # a CFG pair: both halves of x are the same rows -- the convolution once, its output copied (the skip
        # tensor is read with all rows by the last output block)
        half = R // 2
        d0 = skip_dst(0)
        ops.conv3x3_small(x[:half], self.conv_in[0], self.conv_in[1], d0[:half])
        ops.cast(d0[:half], d0[half:])
      else:
        ops.conv3x3_small(x, self.conv_in[0], self.conv_in[1], skip_dst(0))
    elif kind == "in":
      cur, dst = skip_dst(idx), skip_dst(idx + 1)
      blk = self.in_blocks[idx]
      if blk[0] == "down":
        self._conv_deferred(cur, blk[1], dst, bias=blk[2], stride=2)     # (flushes first: it reads cur)
      else:
        _, r, st = blk
        if st is None:
          self._res(r, cur, tall, dst)
        elif idx == 0 and env.get("pair") and rows is None:
          # the CFG pair's common prefix: ResBlock + the transformer block's head on the first R/2 rows, with the
          # launch plans of THAT row count; the block's tail (from the first cross-attention on) on all rows
          half = R // 2
          self._flush()
          with ops.plan_scope(half, x.shape[1], dt):
            tmp, tmp_full = self._pair_buf("blk_r", (half,) + tuple(dst.shape[1:3]) + (r.cout,), dt, True)
            self._res(r, cur[:half], tall if tall.shape[0] == 1 else tall[:half], tmp)
            self._st(st, tmp, dst, pair=True, x_full=tmp_full)
        else:
          tmp = B_.get("blk_r", (R,) + tuple(dst.shape[1:3]) + (r.cout,), dt)
          self._res(r, cur, tall, tmp)
          self._st(st, tmp, dst)
    elif kind == "mid":
      cur = skip_dst(n_in)
      r1, stm, r2 = self.mid
      shp = (R,) + tuple(cur.shape[1:3]) + (r1.cout,)
      m1 = self._res(r1, cur, tall, B_.get("blk_r", shp, dt))
      m2 = self._st(stm, m1, B_.get("blk_s", shp, dt))
      self._res(r2, m2, tall, sl(cats[0])[..., :r2.cout])
    elif kind == "out":
      j = idx
      r, st, up = self.out_blocks[j]
      xin = sl(cats[j])
      hh, ww = xin.shape[1], xin.shape[2]
      last = j + 1 == len(self.out_blocks)
      dst = sl(env["final"]) if last else sl(cats[j + 1])[..., :r.cout]
      stages = 1 + (st is not None) + (up is not None)
      o = self._res(r, xin, tall, dst if stages == 1 else B_.get("blk_r", (R, hh, ww, r.cout), dt))
      if st is not None:
        stages_left = 1 if up is not None else 0
        o = self._st(st, o, dst if stages_left == 0 else B_.get("blk_s", (R, hh, ww, r.cout), dt))
      if up is not None:
        self._flush()                                           # it reads o
        ops.conv3x3(o, up[0], dst, bias=up[1], upsample=True)   # unet.py:44-47
    else:
      final = sl(env["final"])
      t0 = B_.get("gn", tuple(final.shape), dt)
      self._gn(final, self.gn_out, GN_EPS_RES, True, t0)
      if self.conv_out_mm is not None:
        ops.conv3x3(t0, self.conv_out_mm, out, bias=self.conv_out[1])
      else:
        ops.conv3x3_small(t0, self.conv_out[0], self.conv_out[1], out)

  def __call__(self, inputs, time, context=None, y=None, training=False):
    """unet.py:118 contract: inputs [R,h,w,4], time int [R], context [R,T,D]."""
    if training:
      raise NotImplementedError("the HIP path is inference only")
    x = torch.as_tensor(inputs, dtype=torch.float32).to(self.device).contiguous()
    t = torch.as_tensor(time).to(torch.int32).to(self.device).contiguous()
    if context is not None:
      self.set_context(torch.as_tensor(context).to(self.device).contiguous())
    return self.forward(x, t_rows=t)
