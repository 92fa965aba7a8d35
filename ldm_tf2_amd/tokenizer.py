"""BERT (uncased) WordPiece tokenisation for the prompt front-end.

Replaces the reference's use of HuggingFace `BertTokenizerFast.from_pretrained(
vocab_dir)` (run_ldm_sampler.py:28-46) for the one thing the sampler needs: the
`input_ids` of a prompt, truncated / padded to `max_length` with [CLS]/[SEP]/[PAD].
Written from the published BERT tokenisation algorithm (basic tokenisation:
clean, lower-case, strip accents, split on punctuation and CJK; then greedy
longest-match-first WordPiece with the "##" continuation prefix).  Only a
`vocab.txt` (one token per line, id = line number) is needed; the reference keeps
it under `bert_model/` and the YAML key `ldm_sampling.vocab_dir` points at it.
"""
from __future__ import annotations

import os
import unicodedata

import numpy as np


def _is_punct(ch):
  cp = ord(ch)
  if (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126):
    return True
  return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
  return ((0x4E00 <= cp <= 0x9FFF) or (0x3400 <= cp <= 0x4DBF) or (0x20000 <= cp <= 0x2A6DF) or
          (0x2A700 <= cp <= 0x2B73F) or (0x2B740 <= cp <= 0x2B81F) or (0x2B820 <= cp <= 0x2CEAF) or
          (0xF900 <= cp <= 0xFAFF) or (0x2F800 <= cp <= 0x2FA1F))


class BertWordPieceTokenizer:

  def __init__(self, vocab_dir_or_file, do_lower_case=True, max_chars_per_word=100):
    path = vocab_dir_or_file
    if os.path.isdir(path):
      path = os.path.join(path, "vocab.txt")
    with open(path, encoding="utf-8") as f:
      tokens = [line.rstrip("\n") for line in f]
    self.vocab = {t: i for i, t in enumerate(tokens)}
    self.lower = do_lower_case
    self.max_chars = max_chars_per_word
    self.cls_id, self.sep_id = self.vocab["[CLS]"], self.vocab["[SEP]"]
    self.pad_id, self.unk_id = self.vocab["[PAD]"], self.vocab["[UNK]"]

  def __len__(self):
    return len(self.vocab)

  # -- basic tokenisation ---------------------------------------------------------
  def _basic(self, text):
    out = []
    for ch in text:
      cp = ord(ch)
      if cp == 0 or cp == 0xFFFD or (unicodedata.category(ch) in ("Cc", "Cf") and ch not in "\t\n\r"):
        continue
      if ch in " \t\n\r" or unicodedata.category(ch) == "Zs":
        out.append(" ")
      elif _is_cjk(cp):
        out.extend([" ", ch, " "])
      else:
        out.append(ch)
    words = []
    for tok in "".join(out).split():
      if self.lower:
        tok = tok.lower()
        tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
      cur = []
      for ch in tok:
        if _is_punct(ch):
          if cur:
            words.append("".join(cur))
            cur = []
          words.append(ch)
        else:
          cur.append(ch)
      if cur:
        words.append("".join(cur))
    return words

  # -- wordpiece ----------------------------------------------------------------------
  def _wordpiece(self, word):
    if len(word) > self.max_chars:
      return [self.unk_id]
    ids, start = [], 0
    while start < len(word):
      end, cur = len(word), None
      while start < end:
        sub = word[start:end]
        if start > 0:
          sub = "##" + sub
        if sub in self.vocab:
          cur = self.vocab[sub]
          break
        end -= 1
      if cur is None:
        return [self.unk_id]
      ids.append(cur)
      start = end
    return ids

  def encode(self, text, max_length=77):
    """[CLS] pieces... [SEP], truncated to max_length, padded with [PAD]."""
    ids = []
    for w in self._basic(text):
      ids.extend(self._wordpiece(w))
    ids = ids[:max(0, max_length - 2)]
    ids = [self.cls_id] + ids + [self.sep_id]
    ids = ids + [self.pad_id] * (max_length - len(ids))
    return np.asarray(ids, dtype=np.int64)


def get_token_ids(prompt, batch_size, vocab_dir, max_length=77):
  """run_ldm_sampler.py:28-46: int64 [2B, max_length]; rows 0..B-1 = the empty
  prompt (unconditional), rows B..2B-1 = the prompt."""
  tok = BertWordPieceTokenizer(vocab_dir)
  cond = tok.encode(prompt, max_length)[None]
  uncond = tok.encode("", max_length)[None]
  return np.concatenate([np.tile(uncond, (batch_size, 1)), np.tile(cond, (batch_size, 1))], axis=0)
