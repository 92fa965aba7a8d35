"""End-to-end harness: `python -m ldm_tf2_amd.run_ldm_sampler` on a
YAML with the reference's sections -> images.npy, against the oracle pipeline
(tokenise -> text-encode -> DDIM loop -> decode -> per-image min-max -> uint8).

The vocabulary is a small synthetic WordPiece vocab written by the test (the reference's
bert_model/vocab.txt does not travel to the GPU box); token-id pins against the real BERT vocab
are CPU tests (tests/test_host_cpu.py)."""
import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

from ldm_tf2_amd import weights as Wt  # noqa: E402
from oracle import ldm_oracle as O  # noqa: E402

UNET = dict(model_channels=64, out_channels=4, num_blocks=2, attention_resolutions=[4, 2, 1], dropout_rate=0.1,
            channel_mult=[1, 2, 4, 4], num_heads=8)
TXT = dict(vocab_size=200, encoder_stack_size=2, hidden_size=128, num_heads=4, size_per_head=32,
           max_seq_len=77, filter_size=256, dropout_rate=0.1)
KL = dict(latent_channels=4, channels=64, num_blocks=2, attention_resolutions=[], dropout_rate=0.,
          multipliers=[1, 2, 4, 4], resample_with_conv=True)
LDM = dict(num_steps=1000, beta_start=0.00085, beta_end=0.012, v_posterior=0., scale_factor=0.18215, eta=0.,
           num_ddim_steps=10)
PROMPT = "a painting of a virus monster playing guitar"


def _write_inputs(tmp_path, progressive=False):
  words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "a", "painting", "of", "virus", "monster", "play", "##ing",
           "guitar", "gui", "##tar", "the", ","]
  words += [f"tok{i}" for i in range(200 - len(words))]
  (tmp_path / "vocab.txt").write_text("\n".join(words) + "\n", encoding="utf-8")
  cfg = {
      "ldm_sampling": {"autoencoder_type": "kl", "latent_shape": [2, 16, 16, 4], "guidance_scale": 5.0,
                       "text_prompt": PROMPT, "vocab_dir": str(tmp_path), "sample_save_progress": progressive},
      "pre_ckpt_paths": {"cond_stage_model": None, "unet": None, "autoencoder": None},
      "cond_stage_model": TXT, "autoencoder_kl": KL, "unet": UNET, "ldm": LDM,
  }
  path = tmp_path / "config.yaml"
  path.write_text(yaml.safe_dump(cfg))
  return path


def _oracle_images(tmp_path, seed):
  from ldm_tf2_amd.model_runners import normal_latents
  from ldm_tf2_amd.tokenizer import get_token_ids
  ids = get_token_ids(PROMPT, 2, str(tmp_path), 77)
  assert ids.shape == (4, 77) and ids[0, 0] == 2 and ids[0, 1] == 3 and (ids[0, 2:] == 0).all()   # [CLS][SEP][PAD]..
  assert ids[2, 1] == 4 and (ids[2] == 10).sum() == 1                                              # "a", one "##ing"
  man_u = Wt.unet_manifest(context_dim=TXT["hidden_size"], **UNET)
  w = {"unet": Wt.init_weights(man_u, seed=2, scope="unet"),
       "cond_stage_model": Wt.init_weights(Wt.transformer_manifest(**TXT), seed=2, scope="cond_stage_model"),
       "autoencoder": Wt.init_weights(Wt.decoder_manifest(**KL), seed=2, scope="autoencoder")}
  x_T = normal_latents(seed, 0, 2, (16, 16, 4))
  return ids, w, x_T


def test_cli_writes_the_reference_images_file(dev, tmp_path, monkeypatch):
  from ldm_tf2_amd import run_ldm_sampler as R
  cfg = _write_inputs(tmp_path)
  out = tmp_path / "images.npy"
  R.main(["--config_path", str(cfg), "--dtype", "f32", "--seed", "7", "--out", str(out)])
  got = np.load(out)
  assert got.dtype == np.uint8 and got.shape == (2, 128, 128, 3)            # run_ldm_sampler.py:99
  ids, w, x_T = _oracle_images(tmp_path, 7)
  ref = O.tensor_to_image(O.ddim_p_sample_loop(ids, x_T, w, LDM, guidance_scale=5.0))
  ref = np.asarray(ref)
  diff = np.abs(got.astype(np.int16) - ref.astype(np.int16))
  print("uint8 images: equal %.4f, max diff %d" % ((diff == 0).mean(), diff.max()))
  assert diff.max() <= 1 and (diff == 0).mean() > 0.99                      # truncation boundary cases only
  for i in range(2):                                                         # per-image min-max (:18-25)
    assert got[i].min() == 0 and got[i].max() >= 254


def test_cli_progressive_branch(dev, tmp_path, monkeypatch):
  from ldm_tf2_amd import run_ldm_sampler as R
  cfg = _write_inputs(tmp_path, progressive=True)
  monkeypatch.chdir(tmp_path)                                                # it writes *.npy into the cwd (:91-94)
  R.main(["--config_path", str(cfg), "--dtype", "f32", "--seed", "7"])
  sp, px = np.load(tmp_path / "sample_prog.npy"), np.load(tmp_path / "pred_x0_prog.npy")
  assert sp.dtype == np.uint8 and sp.shape == (2, 2, 128, 128, 3) and px.shape == sp.shape   # 10 steps / record_freq 5
  # the last recorded pred_x0 is what the last DDIM step predicts; its image is not constant
  assert px[:, -1].std() > 1.0
