"""Launch-plan table of ldm_gemm (host logic, no GPU): key format, JSON loading, candidate
enumeration, and that every packaged plan names a tile the library knows and a legal split."""
import ctypes as C
import glob
import json
import os

import pytest
import torch

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
  before = ops.plan_tables()
  f = tmp_path / "plans.json"
  f.write_text(json.dumps({"about": "test", "config": {"rows": 6, "latent": 24, "dtype": "bf16"},
                           "plans": {key: [6, 1]}}))
  try:
    assert ops.load_plans(str(f)) == (6, 24, ops.BF16)
    assert ops.gemm_plans(6, 24, "bf16")[key] == (6, 1)
    assert ops.gemm_plans() == {}                      # nothing is active outside a scope
    with ops.plan_scope(6, 24, torch.bfloat16):
      assert ops.gemm_plans()[key] == (6, 1)
      q = _params(32768, 320, 2880, conv=1)
      q.H = q.W = 32
      q.stride = 1
      ops.resolve_plan(q)
      assert (q.tile, q.split_k) == (6, 1)
      with ops.plan_scope(6, 24, torch.float32):       # another configuration: its own (empty) table
        assert ops.gemm_plans() == {}
      assert ops.gemm_plans()[key] == (6, 1)
      ops.set_plan(key, None)
      assert key not in ops.gemm_plans()
    assert ops.gemm_plans() == {}
    # a second file for the same configuration must not silently override a measured plan
    ops.load_plans(str(f))
    g = tmp_path / "other.json"
    g.write_text(json.dumps({"config": {"rows": 6, "latent": 24, "dtype": "bf16"}, "plans": {key: [1, 1]}}))
    with pytest.raises(ValueError, match="conflicts"):
      ops.load_plans(str(g))
    h = tmp_path / "nohdr.json"
    h.write_text(json.dumps({"plans": {key: [1, 1]}}))
    with pytest.raises(ValueError, match="config"):
      ops.load_plans(str(h))
  finally:
    ops.clear_plans()
    ops._load_default_plans()
  assert ops.plan_tables() == before


def test_packaged_tables_one_per_configuration():
  """Each packaged file names its step configuration, no two files share one, and the table the
  U-Net activates for a configuration is exactly that file's (the C2 table used to be
  overridden by the latent-64 one)."""
  d = os.path.join(os.path.dirname(ops.__file__), "plans")
  seen = {}
  for path in sorted(glob.glob(os.path.join(d, "*.json"))):
    j = json.load(open(path))
    c = j["config"]
    ck = (c["rows"], c["latent"], c["dtype"])
    assert ck not in seen, f"{path} and {seen[ck]} both claim configuration {ck}"
    seen[ck] = path
    want = {k: tuple(v) for k, v in j["plans"].items()}
    assert ops.gemm_plans(*ck) == want


def test_candidates_respect_tile_rules():
  c = ops.plan_candidates(32768, 320, 320, 1, ops.ACT_NONE, ops.BF16)
  assert (6, 1) in c and (8, 1) in c and (1, 1) in c and all(s == 1 for _, s in c)     # fills the chip: no split-K
  c = ops.plan_candidates(512, 1280, 11520, 1, ops.ACT_NONE, ops.BF16)                 # 4x4 conv: split-K matters
  assert (1, 12) in c and (2, 12) in c and (6, 4) in c
  c = ops.plan_candidates(32768, 2560, 320, 1, ops.ACT_GEGLU, ops.BF16)
  assert {t for t, _ in c} == {1, 2, 11, 12, 14, 19}                                                # GEGLU: 64-column interleave
  c = ops.plan_candidates(8192, 1536, 640, 1, ops.ACT_NONE, ops.BF16)
  assert all(t < 6 or t in (11, 12, 14, 17, 18, 19) for t, _ in c)                                            # N % 160 != 0
  assert all(t not in (9, 10, 11, 12, 13, 14) for t, _ in ops.plan_candidates(8192, 640, 640, 1, ops.ACT_NONE, ops.F32))   # bf16-only tiles


def test_packaged_plans_are_legal():
  d = os.path.join(os.path.dirname(ops.__file__), "plans")
  files = sorted(glob.glob(os.path.join(d, "*.json")))
  for path in files:
    plans = json.load(open(path))["plans"]
    for key, (tile, split) in plans.items():
      f = dict((k.rstrip("0123456789"), int(k[len(k.rstrip("0123456789")):])) for k in key.split())
      assert tile in (1, 2, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19) and split >= 1, (path, key)
      assert (tile, split) in ops.plan_candidates(f["M"], f["N"], f["K"], f["b"], f["act"], f["dt"], key=key), (path, key)
      # the library accepts the forced pair for planning purposes
      p = _params(f["M"], f["N"], f["K"], conv=f["conv"], act=f["act"], dtype=f["dt"])
      p.tile, p.split_k = tile, split
      p.workspace, p.workspace_bytes = 1, 1 << 30
      t, s = C.c_int(), C.c_int()
      assert lib.ldm_gemm_plan(C.byref(p), C.byref(t), C.byref(s)) == 0
      assert t.value == tile


def test_halo_ring_candidates_follow_the_conv_geometry():
  """Tiles 15 / 16 are only offered where the library runs them: stride-1 convs with W = 16 / 32 whose
  256-row M-tile is whole lines of one image (the plan key carries the geometry)."""
  k32 = "M32768 N320 K2880 b1 conv1 H32 W32 s1 u0 nlp0 act0 dt1 odt1"
  k8 = "M2048 N1280 K11520 b1 conv1 H8 W8 s1 u0 nlp0 act0 dt1 odt1"
  ks2 = "M8192 N320 K2880 b1 conv1 H32 W32 s2 u0 nlp0 act0 dt1 odt1"
  kup = "M32768 N640 K5760 b1 conv1 H16 W16 s1 u1 nlp0 act0 dt1 odt1"
  tiles = lambda key, M, N, K, dt=ops.BF16: {t for t, _ in ops.plan_candidates(M, N, K, 1, 0, dt, key=key)}
  assert 15 in tiles(k32, 32768, 320, 2880) and 16 not in tiles(k32, 32768, 320, 2880)      # 320 = 2 x 160
  assert not {15, 16} & tiles(k8, 2048, 1280, 11520)
  assert not {15, 16} & tiles(ks2, 8192, 320, 2880)
  assert not {15, 16} & tiles(kup, 32768, 640, 5760)
  assert not {15, 16} & tiles(k32.replace("dt1 odt1", "dt0 odt0"), 32768, 320, 2880, ops.F32)
  assert not {15, 16} & tiles(None, 32768, 320, 2880)
  # a second A operand (" x2": the ResBlock shortcut inside the convolution, FF-out + proj_out folded) runs on the
  # implicit-GEMM tiles only: neither the halo-staged nor the persistent tiles are offered for such a key
  kx = "M32768 N320 K3840 b1 conv1 H32 W32 s1 u0 nlp0 act0 dt1 odt1 x2"
  assert not {13, 14, 15, 16} & tiles(kx, 32768, 320, 3840) and 9 in tiles(kx, 32768, 320, 3840)
  kd = "M8192 N640 K3200 b1 conv0 H0 W0 s0 u0 nlp0 act0 dt1 odt1 x2"
  assert not {13, 14} & tiles(kd, 8192, 640, 3200)
