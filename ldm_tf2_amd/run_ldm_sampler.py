"""Sampler harness: YAML -> three models -> prompt tokens -> DDIM loop -> images.npy.

Same I/O contract as the reference CLI (run_ldm_sampler.py:49-99): reads the YAML
sections `ldm_sampling`, `pre_ckpt_paths`, `cond_stage_model`, `autoencoder_kl|vq`,
`unet`, `ldm` (keys == constructor kwargs), tiles the prompt into ids
[uncond x B; cond x B], samples, converts with the per-image min-max rule
(:18-25) and writes uint8 NHWC `images.npy`.

    python -m ldm_tf2_amd.run_ldm_sampler --config_path all_in_one_config.yaml \
        [--dtype bf16|f32] [--seed 0] [--out images.npy]

Checkpoints: the reference restores TF checkpoints with expect_partial(), which
silently leaves random-init weights when nothing matches.
TF checkpoints cannot be read here; `pre_ckpt_paths` entries that point at an
`.npz` of reference-layout arrays (weights.py names) are loaded, and an extra key
`pre_ckpt_paths.compvis` naming the original PyTorch checkpoint loads all three models
through checkpoint.py; anything else falls back to seeded random init with a warning --
the same observable behaviour.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch
import yaml

from . import ops
from .autoencoder import AutoencoderKL, AutoencoderVQ
from .model_runners import LatentDiffusionModelSampler
from .tokenizer import get_token_ids
from .transformer import TransformerModel
from .unet import UNet


def tensor_to_image(images):
  """run_ldm_sampler.py:18-25 on the device: per image (x-min)/(max-min)*255 ->
  uint8 (truncation).  Returns a NumPy uint8 array."""
  x = images.contiguous()
  out = torch.empty(tuple(x.shape), dtype=torch.uint8, device=x.device)
  ops.minmax_u8(x, out)
  return out.cpu().numpy()


_preloaded = {}


def _load_weights(path, what):
  if what in _preloaded:
    return _preloaded.pop(what)
  if path and os.path.isfile(path) and path.endswith(".npz"):
    return dict(np.load(path))
  print(f"[WARN] no loadable checkpoint for {what} at {path!r}: using random-init weights "
        "(the reference's expect_partial() behaviour)", file=sys.stderr)
  return None


def compvis_manifest_configs(config):
  """Manifest kwargs of the three models as `build_from_config` will build them (what a CompVis
  checkpoint must be mapped onto): U-Net with the text model's hidden size as context width, the
  autoencoder section the sampler selects, for VQ with the run-time latent size and codebook."""
  kind = config["ldm_sampling"]["autoencoder_type"]
  unet_cfg = dict(config["unet"], context_dim=config["cond_stage_model"]["hidden_size"])
  if kind == "kl":
    ae_cfg = dict(config["autoencoder_kl"], attention_resolutions=())      # KL ignores it (autoencoder.py:339)
  else:
    ae_cfg = dict(config["autoencoder_vq"], latent_size=config["ldm_sampling"]["latent_shape"][1])
  return dict(unet_cfg=unet_cfg, transformer_cfg=dict(config["cond_stage_model"]), autoencoder_cfg=ae_cfg)


def build_from_config(config, dtype=torch.bfloat16, device="cuda:0", seed=2, use_graph=True,
                      verbose=True):
  ck = dict(config.get("pre_ckpt_paths", {}))
  kind = config["ldm_sampling"]["autoencoder_type"]
  if kind not in ("kl", "vq"):
    raise NotImplementedError("invalid autoencoder type.")
  latent_size = config["ldm_sampling"]["latent_shape"][1]
  hidden = config["cond_stage_model"]["hidden_size"]
  if ck.get("compvis"):
    # one CompVis PyTorch checkpoint for all three models (checkpoint.py; what the
    # reference reaches through convert_ckpt_pytorch_to_tf2.py + three TF checkpoints).  The
    # sampling path decodes only: the checkpoint's encoder is not loaded (its layout differs
    # between KL, double_z, and VQ); the U-Net's context width is the text model's hidden size;
    # a VQ decoder's attention blocks depend on the latent size it will run at.
    from .checkpoint import from_compvis_state_dict
    sd = torch.load(ck["compvis"], map_location="cpu", weights_only=True)
    sd = {k: v.float().numpy() for k, v in sd.get("state_dict", sd).items() if torch.is_tensor(v)}
    loaded = from_compvis_state_dict(sd, **compvis_manifest_configs(config), with_encoder=False, kl=(kind == "kl"))
    _preloaded.update(loaded)
  transformer = TransformerModel(**config["cond_stage_model"], dtype=dtype, device=device, seed=seed,
                                 weights=_load_weights(ck.get("cond_stage_model"), "cond_stage_model"))
  unet = UNet(**config["unet"], dtype=dtype, device=device, seed=seed,
              context_dim=hidden,
              weights=_load_weights(ck.get("unet"), "unet"))
  if kind == "kl":
    autoencoder = AutoencoderKL(**config["autoencoder_kl"], dtype=dtype, device=device, seed=seed,
                                weights=_load_weights(ck.get("autoencoder"), "autoencoder"))
  elif kind == "vq":
    autoencoder = AutoencoderVQ(**config["autoencoder_vq"], dtype=dtype, device=device, seed=seed,
                                latent_size=latent_size,
                                weights=_load_weights(ck.get("autoencoder"), "autoencoder"))
  return LatentDiffusionModelSampler(unet=unet, autoencoder=autoencoder, cond_stage_model=transformer,
                                     use_graph=use_graph, verbose=verbose, **config["ldm"])


def main(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument("--config_path", required=True)
  ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
  ap.add_argument("--seed", type=int, default=0, help="seed of x_T (and of the noise when eta > 0)")
  ap.add_argument("--out", default="images.npy")
  args = ap.parse_args(argv)
  with open(args.config_path) as f:
    config = yaml.safe_load(f)
  dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
  sampler = build_from_config(config, dtype=dtype)
  samp = config["ldm_sampling"]
  token_ids = get_token_ids(samp["text_prompt"], samp["latent_shape"][0], samp["vocab_dir"],
                            config["cond_stage_model"]["max_seq_len"])
  if samp.get("sample_save_progress"):
    # run_ldm_sampler.py:89-94 (with the reference's unpacking bug fixed: three results)
    _, sample_prog, pred_x0_prog = sampler.ddim_p_sample_loop_progressive(
        token_ids, samp["latent_shape"], samp["guidance_scale"], seed=args.seed)
    for name, t in (("sample_prog.npy", sample_prog), ("pred_x0_prog.npy", pred_x0_prog)):
      print(f"[INFO] Save progressive images to '{name}'...")
      b, r = t.shape[0], t.shape[1]
      # tensor_to_image normalises inputs[i] over everything but the batch axis (:19-22)
      u8 = tensor_to_image(t.reshape(b, r * t.shape[2], t.shape[3], t.shape[4]))
      np.save(name, u8.reshape(tuple(t.shape)))
    return
  images = sampler.ddim_p_sample_loop(token_ids, samp["latent_shape"], samp["guidance_scale"],
                                      seed=args.seed)
  print(f"[INFO] Save generated images to '{args.out}'...")
  np.save(args.out, tensor_to_image(images))


if __name__ == "__main__":
  main()
