"""Pins checkpoint.py / weights.py against the reference's own weight-order specification
(the ordered key lists of convert_ckpt_pytorch_to_tf2.py:23-372 are the one
thing the reference holds that fixes variable order and layout).

Runs in the build container only (the reference does not travel to the GPU box).  Nothing of
the converter is stored here: at test time its four pure list-building functions
(get_transformer_weights / get_unet_weights / get_decoder_weights / get_encoder_weights; they
touch nothing but `sd[...]` and array methods) are cut out of the file with `ast`, executed
against a recording dict of SYMBOLIC arrays (shape + the sequence of .T / .transpose / .squeeze /
.reshape applied), and the recorded (checkpoint key, transform, resulting shape) list is compared,
IN ORDER, with this build's manifests and name rules:
  (i)   the key set equals the union of checkpoint.py's rule keys (per model);
  (ii)  the order equals the manifest order of weights.py (= Keras variable creation order, which
        is what `set_weights(list)` relies on, convert_...:399,:410,:417-429);
  (iii) each key's transform is the rule's transform (conv OIHW->HWIO, Linear .T, 1x1 conv
        squeeze().T, attention split / merge reshapes) and yields the manifest's shape.
"""
import ast
import os

import numpy as np
import pytest

from ldm_tf2_amd import checkpoint as C
from ldm_tf2_amd import weights as W

REF = "/root/reference/convert_ckpt_pytorch_to_tf2.py"
pytestmark = pytest.mark.skipif(not os.path.isfile(REF), reason="reference tree not present (GPU box)")


class Sym:
  """Shape-only stand-in for a checkpoint tensor that records the layout ops applied to it."""

  def __init__(self, key, shape, ops=()):
    self.key, self.shape, self.ops = key, tuple(shape), tuple(ops)

  @property
  def T(self):
    return Sym(self.key, self.shape[::-1], self.ops + (("T",),))

  def transpose(self, *axes):
    axes = axes[0] if len(axes) == 1 and isinstance(axes[0], (tuple, list)) else axes
    return Sym(self.key, [self.shape[a] for a in axes], self.ops + (("transpose", tuple(axes)),))

  def squeeze(self):
    return Sym(self.key, [s for s in self.shape if s != 1], self.ops + (("squeeze",),))

  def reshape(self, *shape):
    shape = list(shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape)
    n = int(np.prod(self.shape))
    if -1 in shape:
      i = shape.index(-1)
      shape[i] = n // int(np.prod([s for s in shape if s != -1]))
    assert int(np.prod(shape)) == n, (self.key, self.shape, shape)
    return Sym(self.key, shape, self.ops + (("reshape", len(shape)),))


def _kind_of_ops(ops):
  names = tuple(o[0] for o in ops)
  if names == ():
    return "id"
  if names == ("transpose",) and ops[0][1] == (2, 3, 1, 0):
    return "conv"
  if names == ("T",):
    return "lin"
  if names == ("squeeze", "T"):
    return "c1"
  if names == ("T", "reshape"):
    return "heads"          # split or merge: told apart by the resulting shape below
  raise AssertionError(f"unrecognised transform {ops}")


def _kind_of_rule(fn):
  if fn is C._ID:
    return "id"
  if fn is C._conv:
    return "conv"
  if fn is C._lin:
    return "lin"
  if fn is C._c1:
    return "c1"
  if getattr(fn, "_kind", None) in ("split", "merge"):
    return "heads"
  raise AssertionError("unknown rule transform")


def _pt_shape(kind, fn, shape):
  """PyTorch-side shape of a variable whose reference-layout shape is `shape`."""
  if kind == "id":
    return shape
  if kind == "conv":                       # HWIO -> OIHW
    return (shape[3], shape[2], shape[0], shape[1])
  if kind == "lin":
    return shape[::-1]
  if kind == "c1":                         # [I, O] -> [O, I, 1, 1]
    return (shape[1], shape[0], 1, 1)
  if fn._kind == "split":                  # [D, H, S] -> [H*S, D]
    return (shape[1] * shape[2], shape[0])
  return (shape[2], shape[0] * shape[1])   # merge [H, S, D] -> [D, H*S]


def _converter_functions():
  tree = ast.parse(open(REF).read())
  want = {"get_transformer_weights", "get_unet_weights", "get_decoder_weights", "get_encoder_weights"}
  fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
  assert {f.name for f in fns} == want
  for f in fns:
    _vet(f)
  # only these four (vetted) defs run, and with two builtins: the reference tree is untrusted content
  ns = {"__builtins__": {"range": range, "str": str}}
  exec(compile(ast.Module(body=fns, type_ignores=[]), REF, "exec"), ns)
  return ns


# What the four list-building functions are made of (checked before anything of them executes): plain
# data flow -- loops, subscripts, list.append, the four layout transforms -- and nothing that could
# reach outside: no imports, no attribute other than the transforms, no call other than range / str /
# the functions' own nested helpers, no decorators, no default arguments, no dunder names.
_NODES = {"Add", "Assign", "Attribute", "BinOp", "Call", "Compare", "Constant", "Continue", "Dict", "Eq", "Expr",
          "For", "FormattedValue", "FunctionDef", "If", "In", "JoinedStr", "List", "Load", "Mult", "Name", "Return",
          "Store", "Subscript", "Tuple", "USub", "UnaryOp", "arg", "arguments", "Sub", "NotEq", "NotIn", "Lt", "Gt",
          "LtE", "GtE", "And", "Or", "BoolOp", "Not", "Slice", "AugAssign", "Pass", "Break"}
_ATTRS = {"append", "extend", "transpose", "T", "reshape", "squeeze"}


def _vet(fn):
  local_defs = {n.name for n in ast.walk(fn) if isinstance(n, ast.FunctionDef)}
  for n in ast.walk(fn):
    kind = type(n).__name__
    assert kind in _NODES, f"{fn.name}: unexpected construct {kind} (line {getattr(n, 'lineno', '?')})"
    if isinstance(n, ast.FunctionDef):
      assert not n.decorator_list and not n.args.defaults and not n.args.kw_defaults and \
          n.args.vararg is None and n.args.kwarg is None, f"{fn.name}: decorators / defaults in {n.name}"
    if isinstance(n, ast.Attribute):
      assert n.attr in _ATTRS, f"{fn.name}: attribute .{n.attr}"
    if isinstance(n, ast.Name):
      assert not n.id.startswith("__"), f"{fn.name}: name {n.id}"
    if isinstance(n, ast.Call) and isinstance(n.func, ast.Name):
      assert n.func.id in {"range", "str"} | local_defs, f"{fn.name}: call to {n.func.id}"


def _recorded(fn, rules, manifest):
  """Calls one converter function on symbolic tensors; returns [(key, kind, final shape)]."""
  theirs = {v[0]: (ours, v[1]) for ours, v in rules.items()}

  class Rec(dict):
    def __missing__(self, key):
      assert key in theirs, f"the converter reads '{key}', which no checkpoint.py rule maps"
      ours, tf = theirs[key]
      return Sym(key, _pt_shape(_kind_of_rule(tf), tf, tuple(manifest[ours][0])))

  out = fn(Rec())
  return [(s.key, _kind_of_ops(s.ops), s.shape) for s in out], theirs


def _check(fn_name, rules, manifest, extra=()):
  rec, theirs = _recorded(_converter_functions()[fn_name], rules, manifest)
  rec = list(rec)
  for key in extra:                                 # set outside the list functions (convert_...:417-425)
    ours, tf = theirs[key]
    rec.append((key, _kind_of_rule(tf), tuple(manifest[ours][0])))
  keys = [k for k, _, _ in rec]
  assert len(keys) == len(set(keys))
  assert set(keys) == set(theirs), (sorted(set(theirs) - set(keys))[:3], sorted(set(keys) - set(theirs))[:3])
  for key, kind, shape in rec:
    ours, tf = theirs[key]
    assert kind == _kind_of_rule(tf), (key, kind)
    assert tuple(shape) == tuple(manifest[ours][0]), (key, shape, manifest[ours][0])
  return [theirs[k][0] for k in keys]


def test_unet_keys_order_and_layouts():
  m = W.unet_manifest()
  order = _check("get_unet_weights", C.unet_rules(m), m)
  assert order == list(m), next((a, b) for a, b in zip(order, list(m)) if a != b)
  assert len(order) == len(m) and sum(int(np.prod(s[0])) for s in m.values()) == 872_300_484   # README.md:33


def test_transformer_keys_order_and_layouts():
  m = W.transformer_manifest()
  order = _check("get_transformer_weights", C.transformer_rules(m), m)
  # Keras creates the two embeddings first (transformer.py:249-250); the converter appends them
  # last because set_weights() follows `model.weights`, where the sub-layers built inside call()
  # come ... in the order the converter lists.  The manifest must agree with that list.
  assert order == list(m), next((a, b) for a, b in zip(order, list(m)) if a != b)


def test_decoder_and_encoder_keys_order_and_layouts():
  dm = W.decoder_manifest()
  em = W.encoder_manifest()
  full = dict(dm)
  full.update(em)
  rules = dict(C.decoder_rules(dm))
  rules.update(C.encoder_rules(em))
  dec_rules = {k: v for k, v in rules.items() if k.startswith("decoder/")}
  enc_rules = {k: v for k, v in rules.items() if k.startswith("encoder/")}
  d_order = _check("get_decoder_weights", dec_rules, full)
  assert d_order == [k for k in dm if k.startswith("decoder/")]
  e_order = _check("get_encoder_weights", enc_rules, full)
  assert e_order == [k for k in em if k.startswith("encoder/")]
  # quant / post-quant 1x1 convs are set directly (convert_...:417-425): squeeze().T
  for ours, theirs in (("post_quant_conv/kernel", "first_stage_model.post_quant_conv.weight"),
                       ("quant_conv/kernel", "first_stage_model.quant_conv.weight")):
    assert rules[ours][0] == theirs and rules[ours][1] is C._c1
