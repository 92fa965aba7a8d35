#!/usr/bin/env python3
"""Headline benchmark: images/sec of latent-diffusion sampling on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one batch: text-encode -> 200 DDIM
steps (each one U-Net forward on 2B rows + CFG/DDIM update) -> KL decode (+ one RCCL
all-gather of the decoded images when N > 1).  Workload = BASELINE.json configs[2]
(the configuration the metric is quoted on): txt2img-f8 1.45B, B=16 per GPU, latent
32x32x4, 200 DDIM steps, CFG 5, bf16 U-Net with the f32 scheduler, f32 text encoder + KL
decoder (the all-bf16 pass is reported beside it); random-init weights, synthetic x_T and random BERT token ids (no network:
no checkpoints, no datasets).  Weak scaling: every GPU samples its own 16 images.

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  roofline     : the MFMA GEMM/implicit-conv kernel family (dominant kernel):
                 algorithmic FLOPs per U-Net step in that family / its summed launch
                 time per step = (replay time of the captured step) - (replay time of the
                 same step captured without the family's launches), HIP events on the
                 launch stream over 10 replays each, after the timed region.
  cpu_baseline : the CPU oracle (a port of the reference arithmetic; TensorFlow is not
                 installable here) timed on the host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work (SURVEY.md section 8d / BASELINE.md section 2), GFLOP per U-Net row @32x32
GF_UNET_ROW = {32: 182.48, 64: 809.54}
GF_CONV_ROW = {32: 100.1, 64: 400.4}
GF_GEMM_ROW = {32: 73.5, 64: 279.3}   # 64: token GEMMs x4, context K/V unchanged
GF_CTX_KV_ROW = 4.9          # cross-attention K/V GEMMs, hoisted out of the step (step-invariant)
# the CFG pair's common prefix (UNet.forward(paired_rows=True)): the first ResBlock's two 320 -> 320 convolutions and
# the first transformer block's proj_in / q / k / v projections are evaluated for HALF the rows -- the family's
# executed work is smaller by this much per skipped row (the roofline leg prices executed, not algorithmic, FLOPs)
GF_PAIR_PREFIX_ROW = {32: 2 * 1024 * 2 * 9 * 320 * 320 / 1e9 + 4 * 1024 * 2 * 320 * 320 / 1e9}
GF_PAIR_PREFIX_ROW[64] = 4 * GF_PAIR_PREFIX_ROW[32]
GF_DECODE = {32: 623.11, 64: 2518.3}
GF_TEXT_ROW = 77.96
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md: dense MFMA peaks

FULL = dict(
    cond_stage_model=dict(vocab_size=30522, encoder_stack_size=32, hidden_size=1280, num_heads=8,
                          size_per_head=64, max_seq_len=77, filter_size=5120, dropout_rate=0.1),
    autoencoder_kl=dict(latent_channels=4, channels=128, num_blocks=2, attention_resolutions=[],
                        dropout_rate=0., multipliers=[1, 2, 4, 4], resample_with_conv=True),
    unet=dict(model_channels=320, out_channels=4, num_blocks=2, attention_resolutions=[4, 2, 1],
              dropout_rate=0.1, channel_mult=[1, 2, 4, 4], num_heads=8),
    ldm=dict(num_steps=1000, beta_start=0.00085, beta_end=0.012, v_posterior=0., scale_factor=0.18215,
             eta=0., num_ddim_steps=200),
)


# BASELINE.json configs -> (batch per GPU, latent, DDIM steps, U-Net dtype)
CONFIGS = {
    # decoder_dtype = text encoder + KL decoder: configs[2] / [3] name only the U-Net as bf16 ("bf16
    # U-Net + fp32 scheduler"), so the headline runs them in f32; the all-bf16 pass is the reported
    # variant (`value_with_bf16_text_and_decoder`)
    "c2": dict(batch_per_gpu=4, latent=32, ddim_steps=50, dtype="f32", decoder_dtype="f32"),
    "c3": dict(batch_per_gpu=16, latent=32, ddim_steps=200, dtype="bf16", decoder_dtype="f32"),
    "c4": dict(batch_per_gpu=8, latent=32, ddim_steps=200, dtype="bf16", decoder_dtype="f32"),
    "c5": dict(batch_per_gpu=4, latent=64, ddim_steps=200, dtype="f32", decoder_dtype="f32"),
    "c5bf16": dict(batch_per_gpu=4, latent=64, ddim_steps=200, dtype="bf16", decoder_dtype="bf16"),
}


def synthetic_token_ids(batch, seed=1, vocab=30522, T=77):
  """uncond rows = [CLS][SEP][PAD]...; cond rows = uniform random ids."""
  uncond = np.array([[101, 102] + [0] * (T - 2)], dtype=np.int64)
  cond = np.random.default_rng(seed).integers(0, vocab, size=(1, T), dtype=np.int64)
  return np.concatenate([np.tile(uncond, (batch, 1)), np.tile(cond, (batch, 1))], 0)


def log(rank, *a):
  if rank == 0:
    print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baseline(weights, latent, n_ddim):
  """CPU oracle (a port of the reference arithmetic: TensorFlow is not installable) on a
  BOUNDED sample: BASELINE configs[0] itself -- run_ldm_sampler's path at latent_shape
  [1,32,32,4], num_ddim_steps=10, random-init full-size weights, fixed seed: text-encode 2 rows,
  10 DDIM steps (one CFG U-Net evaluation each), KL decode -- run once, for real (~15-20 s).
  images/s at the bench's step count = 1 / (n_ddim * measured seconds per DDIM step + text +
  decode).  Threads = the cores this process may run on (affinity), all of them.  Test
  infrastructure used as the checker's clock only -- never part of the measured GPU path."""
  from oracle import ldm_oracle as O
  try:
    affinity = len(os.sched_getaffinity(0))
  except AttributeError:
    affinity = os.cpu_count() or 1
  # threads = the CPU share this process really has: the cgroup quota when one is set (a GPU box
  # hands one GPU's job 16 CPUs' worth of time while its affinity mask shows all 256 hardware
  # threads -- 256 torch threads on a 16-CPU quota ran this sample 200x slower), else the
  # affinity count, never more than 64 (the oracle's GEMMs stop scaling there)
  quota = None
  try:
    with open("/sys/fs/cgroup/cpu.max") as f:
      q, per = f.read().split()[:2]
      if q != "max":
        quota = max(1, int(float(q) / float(per) + 0.5))
  except (OSError, ValueError):
    pass
  if quota is None:
    try:                                                       # cgroup v1
      q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
      per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
      if q > 0:
        quota = max(1, int(q / per + 0.5))
    except (OSError, ValueError):
      pass
  share = quota if quota is not None else min(affinity, 16)
  cores = max(1, int(os.environ.get("LDM_CPU_BASELINE_THREADS", min(share, affinity, 64))))
  torch.set_num_threads(cores)
  cpu_model = "unknown CPU"
  try:
    with open("/proc/cpuinfo") as f:
      for line in f:
        if line.startswith("model name"):
          cpu_model = line.split(":", 1)[1].strip()
          break
  except OSError:
    pass
  g = np.random.default_rng(0)
  ids = synthetic_token_ids(1)
  x_T = g.standard_normal((1, latent, latent, 4)).astype(np.float32)
  n_cpu = 10
  ldm = dict(FULL["ldm"], num_ddim_steps=n_cpu)
  sched = O.make_schedule(ldm["num_steps"], ldm["beta_start"], ldm["beta_end"], 0., n_cpu)
  with torch.no_grad():
    t0 = time.perf_counter()
    ctx = O.text_encoder(ids, weights["cond_stage_model"])
    t_text = time.perf_counter() - t0
    log(0, f"cpu oracle: text encoder {t_text:.1f}s ({cores} threads, affinity {affinity})")
    xt = torch.from_numpy(x_T)
    t0 = time.perf_counter()
    budget_hit = False
    done = 0
    for index in range(n_cpu - 1, -1, -1):
      xt, _, _ = O.ddim_sample(xt, ctx, index, sched, weights["unet"], 5.0, None, clip_denoised=False)
      done += 1
      if time.perf_counter() - t0 > 40.0:       # slow host: stop early, still a measured per-step time
        budget_hit = True
        break
    t_loop = time.perf_counter() - t0
    t_step = t_loop / done
    log(0, f"cpu oracle: {done} DDIM steps in {t_loop:.1f}s")
    t0 = time.perf_counter()
    O.decoder_forward(xt / ldm["scale_factor"], weights["autoencoder"])
    t_dec = time.perf_counter() - t0
    log(0, f"cpu oracle: decode {t_dec:.1f}s")
  per_image = n_ddim * t_step + t_dec + t_text
  c1_images_per_s = 1.0 / (n_cpu * t_step + t_dec + t_text)
  return {
      "value": 1.0 / per_image, "unit": "images/s", "cores": cores, "affinity_cores": affinity,
      "cgroup_cpu_quota": quota,
      "cpu_model": cpu_model, "kind": "port",
      "sample": (f"torch-CPU f32 oracle, {cores} threads: BASELINE configs[0] run once (latent [1,{latent},{latent},4], "
                 f"{done}{' of 10 (40 s budget)' if budget_hit else ''} DDIM steps with CFG = {t_loop:.2f}s, text-encode 2 rows = "
                 f"{t_text:.2f}s, KL decode = {t_dec:.2f}s => {c1_images_per_s:.4f} images/s at 10 steps); value = "
                 f"1/({n_ddim} * {t_step:.3f}s + decode + text)"),
      "ms_per_unet_step": t_step * 1e3, "c1_images_per_s_10_steps": c1_images_per_s,
  }


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=2)
  ap.add_argument("--warmup", type=int, default=1)
  ap.add_argument("--config", default="c3", choices=sorted(CONFIGS),
                  help="BASELINE.json configuration: c3 = configs[2] (the one the metric is quoted on, default), "
                       "c2 = configs[1], c4 = configs[3] (8 per GPU: the multi-GPU scaling point), c5 = configs[4]")
  ap.add_argument("--batch-per-gpu", type=int, default=None)
  ap.add_argument("--ddim-steps", type=int, default=None)
  ap.add_argument("--latent", type=int, default=None)
  ap.add_argument("--dtype", default=None, choices=["bf16", "f32"])
  ap.add_argument("--decoder-dtype", default=None, choices=["bf16", "f32"],
                  help="dtype of the text encoder + KL decoder (default: the configuration's -- f32 for "
                       "c2..c5, which name only the U-Net as bf16)")
  ap.add_argument("--guidance", type=float, default=5.0)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-graph", action="store_true")
  ap.add_argument("--no-variant", action="store_true",
                  help="skip the extra pass with the text encoder / decoder in the other precision")
  ap.add_argument("--no-plans", action="store_true",
                  help="A/B: ldm_gemm's cost-model tile/split choice only (ignore the packaged in-situ plan table)")
  ap.add_argument("--tiny", action="store_true",
                  help="TEST ONLY (tests/test_multirank_gpu.py): a few-MB model of the same architecture so the "
                       "multi-rank path of this script runs in seconds; the JSON line says so and is not a benchmark")
  ap.add_argument("--dump-images", default=None,
                  help="rank 0 saves the gathered uint8 images of the last pass to this .npy file")
  ap.add_argument("--first-sample-index", type=int, default=None,
                  help="global index of this process's first sample (default: rank * batch-per-gpu)")
  args = ap.parse_args()
  preset = CONFIGS[args.config]
  for k, v in preset.items():
    if getattr(args, k) is None:
      setattr(args, k, v)

  from ldm_tf2_amd import distributed as D
  rank, world, local = D.init_from_env()
  if world != args.gpus:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs an MI355X: the sampling path has no CPU fallback")
  torch.cuda.set_device(local)
  dev = torch.device("cuda", local)
  dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
  dec_name = args.decoder_dtype or args.dtype
  dec_dtype = torch.bfloat16 if dec_name == "bf16" else torch.float32

  from ldm_tf2_amd import ops, weights as Wt
  from ldm_tf2_amd.autoencoder import AutoencoderKL
  from ldm_tf2_amd.model_runners import LatentDiffusionModelSampler
  from ldm_tf2_amd.transformer import TransformerModel
  from ldm_tf2_amd.unet import UNet

  if args.no_plans:
    ops.clear_plans()
  t_build = time.perf_counter()
  cfg = FULL
  if args.tiny:
    cfg = dict(
        cond_stage_model=dict(vocab_size=30522, encoder_stack_size=2, hidden_size=128, num_heads=4, size_per_head=32,
                              max_seq_len=77, filter_size=256),
        autoencoder_kl=dict(latent_channels=4, channels=64, num_blocks=2, multipliers=[1, 2, 4, 4]),
        unet=dict(model_channels=64, out_channels=4, num_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8,
                  context_dim=128),
        ldm=FULL["ldm"])
  w = {
      "unet": Wt.init_weights(Wt.unet_manifest(**cfg["unet"]), seed=2, scope="unet"),
      "cond_stage_model": Wt.init_weights(Wt.transformer_manifest(**cfg["cond_stage_model"]), seed=2,
                                          scope="cond_stage_model"),
      "autoencoder": Wt.init_weights(Wt.decoder_manifest(**cfg["autoencoder_kl"]), seed=2,
                                     scope="autoencoder"),
  }
  log(rank, f"weights generated in {time.perf_counter() - t_build:.1f}s")
  unet = UNet(**cfg["unet"], weights=w["unet"], dtype=dtype, device=dev)
  txt = TransformerModel(**cfg["cond_stage_model"], weights=w["cond_stage_model"], dtype=dec_dtype, device=dev)
  ae = AutoencoderKL(**cfg["autoencoder_kl"], weights=w["autoencoder"], dtype=dec_dtype, device=dev)
  ldm = dict(cfg["ldm"], num_ddim_steps=args.ddim_steps)
  sampler = LatentDiffusionModelSampler(unet, ae, txt, use_graph=not args.no_graph, verbose=False, **ldm)
  keep_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
  if not keep_cpu:
    w = None
  log(rank, f"models on device in {time.perf_counter() - t_build:.1f}s")

  B = args.batch_per_gpu
  shape = [B, args.latent, args.latent, 4]
  ids = synthetic_token_ids(B)
  first, _ = D.shard_range(rank, B)
  if args.first_sample_index is not None:
    first = args.first_sample_index

  u8 = torch.empty(B, 8 * args.latent, 8 * args.latent, 3, dtype=torch.uint8, device=dev)
  mm_scratch = torch.empty(B * 128, dtype=torch.float32, device=dev)

  def one_pass(smp=None):
    """text-encode -> N DDIM steps -> KL decode -> per-image min-max to uint8 (run_ldm_sampler.py:18-25)
    -> ONE all-gather of the uint8 images (4x less payload than gathering float32)."""
    images = (smp or sampler).ddim_p_sample_loop(ids, shape, guidance_scale=args.guidance, seed=0,
                                                 first_sample_index=first)
    ops.minmax_u8(images.contiguous(), u8, scratch=mm_scratch)
    return D.all_gather_images(u8), images

  for _ in range(args.warmup):
    one_pass()
  D.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  loop_ms = []
  for _ in range(args.steps):
    out, last_f32 = one_pass()
    loop_ms.append(sampler._loop_events)
  D.barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  elapsed = D.max_over_ranks(elapsed, dev)
  ms_unet_step = float(np.mean([a.elapsed_time(b) / n for a, b, n in loop_ms]))
  assert tuple(out.shape) == (world * B, 8 * args.latent, 8 * args.latent, 3) and out.dtype == torch.uint8
  assert bool(torch.isfinite(last_f32.float()).all()), "non-finite images"
  loop_ms_ranks = D.gather_floats(ms_unet_step, dev)
  # the same pass with the text encoder + decoder in the OTHER precision, beside the headline
  # (one warm-up + one timed pass; reported, never the headline)
  other_name = "f32" if dec_name == "bf16" else "bf16"
  other_dtype = torch.float32 if other_name == "f32" else torch.bfloat16
  other_value = None
  if world == 1 and not args.no_variant:
    w2 = w if w is not None else {
        "cond_stage_model": Wt.init_weights(Wt.transformer_manifest(**cfg["cond_stage_model"]), seed=2, scope="cond_stage_model"),
        "autoencoder": Wt.init_weights(Wt.decoder_manifest(**cfg["autoencoder_kl"]), seed=2, scope="autoencoder")}
    txt2 = TransformerModel(**cfg["cond_stage_model"], weights=w2["cond_stage_model"], dtype=other_dtype, device=dev)
    ae2 = AutoencoderKL(**cfg["autoencoder_kl"], weights=w2["autoencoder"], dtype=other_dtype, device=dev)
    s2 = LatentDiffusionModelSampler(unet, ae2, txt2, use_graph=not args.no_graph, verbose=False, **ldm)
    one_pass(s2)
    torch.cuda.synchronize()
    tv = time.perf_counter()
    out2, f32_2 = one_pass(s2)
    torch.cuda.synchronize()
    other_value = B / (time.perf_counter() - tv)
    assert tuple(out2.shape) == tuple(out.shape) and bool(torch.isfinite(f32_2.float()).all()), \
        f"non-finite images in the {other_name} text-encoder / decoder variant"
    del s2, txt2, ae2, out2, f32_2

  # ---- the MFMA GEMM/conv family's share of a step -------------------------------------
  # Two captured HIP graphs of the SAME step, one with every ldm_gemm launch left out
  # (tools.gemm_hooks.skip_gemms), each timed with HIP events over 10 back-to-back replays on the launch
  # stream: family time = full - without.  In a graph the kernels run back to back (their
  # summed rocprof durations equal the replay time), so this agrees with the rocprofv3
  # kernel-trace average in profiles/ and carries no per-launch event overhead.  The older
  # per-launch event brackets (eager step) are kept as `ms_per_unet_step_event_brackets`.
  R = 2 * B
  sampler._index_dev.fill_(args.ddim_steps - 1)

  from tools.gemm_hooks import skip_gemms, time_gemms

  def captured_step(skip_counter):
    import contextlib
    with (skip_gemms(skip_counter) if skip_counter is not None else contextlib.nullcontext()):
      g = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g, capture_error_mode="thread_local"):   # RCCL watchdog thread may be live
        sampler._step(args.guidance, False, None, dec_index=False)
    return g

  def replay_ms(g, reps=10):
    for _ in range(3):
      g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps

  skipped = [0]
  g_full, g_rest = captured_step(None), captured_step(skipped)
  t_full, t_rest = replay_ms(g_full), replay_ms(g_rest)
  gemm_ms = t_full - t_rest
  n_launches = skipped[0]
  del g_full, g_rest
  timers = []
  if sampler._graph is not None:
    for _ in range(3):          # keep the GPU busy while the host enqueues the bracketed step
      sampler._graph.replay()
  sampler._index_dev.fill_(args.ddim_steps - 1)
  with time_gemms(timers):
    sampler._step(args.guidance, False, None, dec_index=False)
    torch.cuda.synchronize()
  bracket_ms = sum(rec[0].elapsed_time(rec[1]) for rec in timers)
  lat = args.latent
  gf_family = ((GF_CONV_ROW.get(lat, 0) + GF_GEMM_ROW.get(lat, 0) - GF_CTX_KV_ROW) * R
               if lat in GF_CONV_ROW and not args.tiny else None)
  gf_pair_skipped = GF_PAIR_PREFIX_ROW.get(lat, 0.0) * (R // 2) if getattr(unet, "_shared_prefix", False) else 0.0
  if gf_family:
    gf_family -= gf_pair_skipped
  roofline = None
  if gf_family:
    achieved = gf_family / gemm_ms          # GFLOP / ms = TFLOP/s
    peak = PEAK_TFLOPS[args.dtype]
    # HBM-side bytes of the family per U-Net step from the committed PMC passes (same workload;
    # separate rocprofv3 --pmc runs, gfx950 FETCH_SIZE correction applied by tools/pmc_family.py)
    traffic, traffic_src = None, ""
    tjs = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
    tj = os.path.join(ROOT, "profiles", tjs[-1]) if tjs else ""     # the latest round's PMC passes
    if args.dtype == "bf16" and lat == 32 and B == 16 and os.path.isfile(tj):
      tjd = json.load(open(tj))
      traffic = tjd.get("hbm_bytes_per_eval")
      traffic_src = f"; measured at commit {tjd['measured_at_commit']} on {tjd.get('measured_on', 'a pool box')}" if tjd.get("measured_at_commit") else ""
    roofline = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic,
                "traffic_unit": ("HBM bytes per U-Net step for this kernel family, from the committed PMC passes of the same "
                                 f"workload (profiles/{os.path.basename(tj) if tj else '-'}; tools/profile_round.sh reproduces them)"
                                 + (traffic_src if traffic is not None else "")),
                "kernel": ("every launch of ldm_gemm: gemm_kernel<T,BM,BN,WM,WN,MODE,MF,ST,NS> + gemm3_kernel<TN,MODE,EPI> (Dense/1x1/"
                           "projection GEMMs incl. the LayerNorm-folded ones + implicit-GEMM 3x3 convs) + their split-K reduces (plain "
                           "launches, and the extra time of the GroupNorm launches that complete a deferred reduce) + the row-panel "
                           "launches that chain the transformer block's products (st_tail_kernel: ldm_st_block / ldm_ffn_geglu; the "
                           "cross-attention inside ldm_st_block is timed with it, its FLOPs are not counted)"),
                "launches_per_unet_step": n_launches, "ms_per_unet_step_in_kernel": gemm_ms,
                "avg_launch_us": gemm_ms * 1e3 / max(n_launches, 1),
                "ms_unet_step_graph_full": t_full, "ms_unet_step_graph_without_family": t_rest,
                "ms_per_unet_step_event_brackets": bracket_ms,
                "algorithmic_gflop_per_unet_step": gf_family,
                "gflop_not_executed_cfg_pair_prefix": gf_pair_skipped}

  if rank == 0 and args.dump_images:
    np.save(args.dump_images, out.cpu().numpy())
  if rank == 0:
    total_images = world * B * args.steps
    key = (B, lat, args.ddim_steps, args.dtype)
    which = "TEST tiny model, not a benchmark" if args.tiny else {(16, 32, 200, "bf16"): "BASELINE configs[2]", (4, 32, 50, "f32"): "BASELINE configs[1]",
             (8, 32, 200, "bf16"): "BASELINE configs[3]", (4, 64, 200, "f32"): "BASELINE configs[4]"}.get(
                 key, "non-BASELINE variant")
    gf_unet = None if args.tiny else GF_UNET_ROW.get(lat)
    value = total_images / elapsed
    res = {
        "metric": "images/sec (256x256, 200 DDIM steps, CFG=5)" if (lat == 32 and args.ddim_steps == 200)
                  else f"images/sec ({8 * lat}x{8 * lat}, {args.ddim_steps} DDIM steps, CFG={args.guidance:g})",
        "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": (f"txt2img-f8 1.45B LDM ({which}): B={B}/GPU, latent {lat}x{lat}x4, "
                                f"{args.ddim_steps} DDIM steps, CFG {args.guidance:g}, {args.dtype} U-Net, {dec_name} text "
                                "encoder + KL decoder, f32 scheduler, random-init weights, synthetic x_T + random BERT ids"),
                   "batch_per_gpu": B, "global_batch": world * B, "ddim_steps": args.ddim_steps,
                   "latent": [lat, lat, 4],
                   "parallelism": f"replicas x{world}, one all-gather of the uint8 images ({B * 64 * lat * lat * 3} B per rank)"},
        "world_size": D.world_size(), "backend": D.backend_name(),
        "ms_per_unet_step_per_rank": loop_ms_ranks,
        f"value_with_{other_name}_text_and_decoder": other_value,
        "ms_per_unet_step": ms_unet_step,
        "unet_tflops": gf_unet * R / ms_unet_step if gf_unet else None,
        "hip_graph": not args.no_graph, "gemm_plan_table_entries": len(ops.gemm_plans(2 * B, lat, args.dtype)),
        "roofline": roofline,
    }
    if keep_cpu:
      log(rank, "timing the CPU oracle (bounded sample)...")
      res["cpu_baseline"] = cpu_baseline(w, lat, args.ddim_steps)
    else:
      res["cpu_baseline"] = None
    print(json.dumps(res), flush=True)
  D.barrier()


if __name__ == "__main__":
  main()
