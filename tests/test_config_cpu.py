"""The reference's own YAML (all_in_one_config.yaml) binds to this build's constructors unchanged:
every key of its `unet`, `cond_stage_model`, `autoencoder_kl`, `autoencoder_vq` and `ldm` sections
is a keyword of the corresponding class (run_ldm_sampler.py:56-83 passes them as **kwargs).
Host-only: signatures are bound, nothing is constructed."""
import inspect
import os

import pytest
import yaml

CFG = "/root/reference/all_in_one_config.yaml"
pytestmark = pytest.mark.skipif(not os.path.isfile(CFG), reason="reference config not present on this box")


def _bind(cls, kwargs):
  sig = inspect.signature(cls.__init__)
  sig.bind_partial(None, **kwargs)          # raises TypeError on an unknown keyword


def test_reference_yaml_sections_bind_to_our_constructors():
  from ldm_tf2_amd.autoencoder import AutoencoderKL, AutoencoderVQ
  from ldm_tf2_amd.model_runners import LatentDiffusionModelSampler
  from ldm_tf2_amd.transformer import TransformerModel
  from ldm_tf2_amd.unet import UNet
  with open(CFG) as f:
    cfg = yaml.safe_load(f)
  for section in ("ldm_sampling", "pre_ckpt_paths", "cond_stage_model", "unet", "ldm"):
    assert section in cfg, section
  _bind(UNet, cfg["unet"])
  _bind(TransformerModel, cfg["cond_stage_model"])
  if "autoencoder_kl" in cfg:
    _bind(AutoencoderKL, cfg["autoencoder_kl"])
  if "autoencoder_vq" in cfg:
    _bind(AutoencoderVQ, cfg["autoencoder_vq"])
  sig = inspect.signature(LatentDiffusionModelSampler.__init__)
  sig.bind_partial(None, unet=None, autoencoder=None, cond_stage_model=None, **cfg["ldm"])
  samp = cfg["ldm_sampling"]
  for key in ("autoencoder_type", "latent_shape", "guidance_scale", "text_prompt", "vocab_dir"):
    assert key in samp, key                                   # the keys run_ldm_sampler.main reads
  assert samp["autoencoder_type"] in ("kl", "vq") and len(samp["latent_shape"]) == 4


def test_reference_model_sizes_from_the_yaml():
  """The YAML's architecture is the 1.45 B txt2img-f8 model: manifests built from its sections have the
  reference's parameter counts (README.md:33)."""
  from ldm_tf2_amd import weights as W
  with open(CFG) as f:
    cfg = yaml.safe_load(f)
  assert W.count_params(W.unet_manifest(**cfg["unet"], context_dim=cfg["cond_stage_model"]["hidden_size"])) == 872300484
  assert W.count_params(W.decoder_manifest(**cfg["autoencoder_kl"])) == 49490199
