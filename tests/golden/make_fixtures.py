#!/usr/bin/env python3
"""Generates the committed fixtures in tests/golden/ (run in the build container,
where /root/reference is mounted; nothing here is needed at test time).

  token_ids.json     : `prompt_ids` / `empty_ids` are the ids the reference hard-codes at
                       convert_ckpt_pytorch_to_tf2.py:384-392 (typed in below as data and
                       re-derived with HuggingFace BertTokenizerFast on the reference's
                       bert_model/vocab.txt, the tokenizer run_ldm_sampler.py:33 uses);
                       `extra` = more prompts tokenised by that same HF tokenizer.
  schedule_kats.json : integer DDIM step tables (closed form of model_runners.py:406-409,
                       last N=50 step pinned to 981 by convert_ckpt_pytorch_to_tf2.py:402)
                       and two alphas_cumprod spot values from the oracle.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

PROMPT = "a virus monster is playing guitar, oil on canvas"
PROMPT_IDS = [101, 1037, 7865, 6071, 2003, 2652, 2858, 1010, 3514, 2006, 10683, 102] + [0] * 65
EMPTY_IDS = [101, 102] + [0] * 75
EXTRA = [
    "A photograph of an astronaut riding a horse!",
    "Zürich's café-crème costs $4.50 (approx.)",
    "unbelievably supercalifragilisticexpialidocious words",
    "two   spaces\tand a tab",
    "painting of 東京 at night",
    "word " * 100,
]


def oracle_pins():
  """oracle_pins.json: outputs of the CPU oracle itself on small seeded problems (weights from
  weights.init_weights, inputs from numpy default_rng: platform-independent).  Not reference
  data -- the reference cannot run here -- but a pin of the checker: an accidental edit of
  oracle/ldm_oracle.py that changes its arithmetic fails tests/test_oracle_pins.py."""
  import numpy as np
  from ldm_tf2_amd import weights as Wt
  from oracle import ldm_oracle as O
  ucfg = dict(model_channels=32, out_channels=4, num_blocks=1, channel_mult=(1, 2), num_heads=4)
  tcfg = dict(vocab_size=100, encoder_stack_size=2, hidden_size=64, num_heads=4, size_per_head=16,
              max_seq_len=16, filter_size=128)
  kcfg = dict(latent_channels=4, channels=32, num_blocks=1, multipliers=(1, 2))
  w = {"unet": Wt.init_weights(Wt.unet_manifest(context_dim=64, **ucfg), seed=7, mode="random", scope="unet"),
       "cond_stage_model": Wt.init_weights(Wt.transformer_manifest(**tcfg), seed=7, mode="random",
                                           scope="cond_stage_model"),
       "autoencoder": Wt.init_weights(Wt.decoder_manifest(**kcfg), seed=7, mode="random", scope="autoencoder")}
  g = np.random.default_rng(11)
  x = g.standard_normal((2, 8, 8, 4)).astype(np.float32)
  ids = g.integers(0, 100, size=(2, 16))
  ctx = O.text_encoder(ids, w["cond_stage_model"], num_heads=4)
  y = O.unet_forward(x, np.array([981, 21], dtype=np.int32), ctx, w["unet"], num_heads=4)
  d = O.decoder_forward(x[:1], w["autoencoder"])
  ldm = dict(num_steps=1000, beta_start=0.00085, beta_end=0.012, v_posterior=0., scale_factor=0.18215, eta=0.,
             num_ddim_steps=10)
  ids4 = np.concatenate([ids[:1], ids[:1], ids[1:], ids[1:]], 0)
  img = O.ddim_p_sample_loop(ids4, x, w, ldm, guidance_scale=5., num_heads=4)
  emb = O.get_time_embedding(np.array([981], dtype=np.int32), 320)
  pin = lambda t: {"shape": list(t.shape), "l2": float(t.double().norm()), "mean": float(t.double().mean()),
                   "head": [float(v) for v in t.flatten()[:8]]}
  out = {"config": {"unet": ucfg, "text": tcfg, "kl": kcfg, "weights_seed": 7, "inputs_seed": 11},
         "text_encoder": pin(ctx), "unet": pin(y), "decoder": pin(d), "ddim_loop_images": pin(img),
         "time_embedding_t981_320": {"cos_0_3": [float(v) for v in emb[0, :3]],
                                     "sin_0_3": [float(v) for v in emb[0, 160:163]]}}
  json.dump(out, open(os.path.join(HERE, "oracle_pins.json"), "w"), indent=1)
  return out


def main():
  if "--oracle-pins-only" in sys.argv:
    oracle_pins()
    return
  os.environ["HF_HUB_OFFLINE"] = "1"
  from transformers import BertTokenizerFast
  tok = BertTokenizerFast.from_pretrained("/root/reference/bert_model")
  enc = lambda s: tok(s, truncation=True, max_length=77, padding="max_length")["input_ids"]
  assert enc(PROMPT) == PROMPT_IDS and enc("") == EMPTY_IDS
  out = {"prompt": PROMPT, "prompt_ids": PROMPT_IDS, "empty_ids": EMPTY_IDS,
         "extra": {s: enc(s) for s in EXTRA}}
  json.dump(out, open(os.path.join(HERE, "token_ids.json"), "w"), ensure_ascii=False, indent=0)

  from oracle import ldm_oracle as O
  g = {"ddim_steps": {}}
  for n in (10, 50, 200):
    steps = [i + 1 for i in range(0, 1000, 1000 // n)]
    assert O.make_schedule(1000, 0.00085, 0.012, 0., n)["ddim_steps"].tolist() == steps
    g["ddim_steps"][str(n)] = steps
  assert g["ddim_steps"]["50"][-1] == 981
  s = O.make_schedule(1000, 0.00085, 0.012, 0., 50)
  g["alphas_cumprod_0"] = float(s["alphas_cumprod"][0])
  g["alphas_cumprod_981"] = float(s["alphas_cumprod"][981])
  json.dump(g, open(os.path.join(HERE, "schedule_kats.json"), "w"))
  oracle_pins()


if __name__ == "__main__":
  main()
