"""Launch-plan table of ldm_gemm (host logic, no GPU): key format, JSON loading, candidate
enumeration, and that every packaged plan names a tile the library knows and a legal split."""
import ctypes as C
import glob
import json
import os

from ldm_tf2_amd import ops
from ldm_tf2_amd._lib import GemmParams, lib


def _params(M, N, K, conv=0, act=0, dtype=ops.BF16):
  p = GemmParams()
  p.M, p.N, p.K, p.batch, p.conv, p.act, p.dtype, p.out_dtype = M, N, K, 1, conv, act, dtype, dtype
  return p


def test_key_and_table_roundtrip(tmp_path):
  p = _params(32768, 320, 2880, conv=1)
  p.H = p.W = 32
  p.stride = 1
  key = ops.plan_key(p)
  assert key == "M32768 N320 K2880 b1 conv1 H32 W32 s1 u0 nlp0 act0 dt%d odt%d" % (ops.BF16, ops.BF16)
  before = ops.gemm_plans()
  f = tmp_path / "plans.json"
  f.write_text(json.dumps({"about": "test", "plans": {key: [6, 1]}}))
  assert ops.load_plans(str(f)) == 1
  assert ops.gemm_plans()[key] == (6, 1)
  ops.set_plan(key, None)
  assert ops.gemm_plans() == before


def test_candidates_respect_tile_rules():
  c = ops.plan_candidates(32768, 320, 320, 1, ops.ACT_NONE, ops.BF16)
  assert (6, 1) in c and (8, 1) in c and (1, 1) in c and all(s == 1 for _, s in c)     # fills the chip: no split-K
  c = ops.plan_candidates(512, 1280, 11520, 1, ops.ACT_NONE, ops.BF16)                 # 4x4 conv: split-K matters
  assert (1, 12) in c and (2, 12) in c and (6, 4) in c
  c = ops.plan_candidates(32768, 2560, 320, 1, ops.ACT_GEGLU, ops.BF16)
  assert {t for t, _ in c} == {1, 2}                                                    # GEGLU: 64-column interleave
  c = ops.plan_candidates(8192, 1536, 640, 1, ops.ACT_NONE, ops.BF16)
  assert all(t < 6 for t, _ in c)                                                       # N % 160 != 0


def test_packaged_plans_are_legal():
  d = os.path.join(os.path.dirname(ops.__file__), "plans")
  files = sorted(glob.glob(os.path.join(d, "*.json")))
  for path in files:
    plans = json.load(open(path))["plans"]
    for key, (tile, split) in plans.items():
      f = dict((k.rstrip("0123456789"), int(k[len(k.rstrip("0123456789")):])) for k in key.split())
      assert tile in (1, 2, 3, 4, 6, 7, 8) and split >= 1, (path, key)
      assert (tile, split) in ops.plan_candidates(f["M"], f["N"], f["K"], f["b"], f["act"], f["dt"]), (path, key)
      # the library accepts the forced pair for planning purposes
      p = _params(f["M"], f["N"], f["K"], conv=f["conv"], act=f["act"], dtype=f["dt"])
      p.tile, p.split_k = tile, split
      p.workspace, p.workspace_bytes = 1, 1 << 30
      t, s = C.c_int(), C.c_int()
      assert lib.ldm_gemm_plan(C.byref(p), C.byref(t), C.byref(s)) == 0
      assert t.value == tile
