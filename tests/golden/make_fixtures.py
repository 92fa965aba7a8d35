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


def main():
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


if __name__ == "__main__":
  main()
