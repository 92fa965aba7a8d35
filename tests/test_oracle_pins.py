"""Pins of the checker itself: the CPU oracle's outputs on small seeded problems against the
committed tests/golden/oracle_pins.json (generator: tests/golden/make_fixtures.py
--oracle-pins-only), plus the timestep-embedding KAT of SURVEY.md A11.  An edit of
oracle/ldm_oracle.py that changes its arithmetic fails here before it can mask a GPU regression."""
import importlib.util
import json
import os

import numpy as np

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_spec = importlib.util.spec_from_file_location("make_fixtures", os.path.join(_GOLDEN, "make_fixtures.py"))
MF = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(MF)
PINS = os.path.join(_GOLDEN, "oracle_pins.json")


def _close(a, b, rtol):
  return abs(a - b) <= rtol * max(abs(b), 1e-6)


def test_oracle_matches_its_committed_pins(tmp_path, monkeypatch):
  want = json.load(open(PINS))
  monkeypatch.setattr(MF, "HERE", str(tmp_path))            # regenerate into a scratch dir
  got = MF.oracle_pins()
  for key in ("text_encoder", "unet", "decoder", "ddim_loop_images"):
    w, g = want[key], got[key]
    assert g["shape"] == w["shape"], key
    assert _close(g["l2"], w["l2"], 2e-5), (key, g["l2"], w["l2"])          # float32 CPU kernels: thread-count noise only
    assert abs(g["mean"] - w["mean"]) <= 2e-5 * max(1.0, abs(w["l2"])), key
    assert np.allclose(g["head"], w["head"], rtol=2e-4, atol=2e-5), key


def test_time_embedding_kat():
  """SURVEY.md A11 (NumPy emulation of unet.py:401-422): t=981, 320 channels, cos first."""
  want = json.load(open(PINS))["time_embedding_t981_320"]
  assert np.allclose(want["cos_0_3"], [0.67996, -0.79843, 0.57811], atol=1e-4)      # SURVEY quotes 5 digits
  assert np.allclose(want["sin_0_3"], [0.73325, 0.60209, 0.81596], atol=1e-4)
