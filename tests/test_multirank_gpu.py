"""The multi-rank path of bench.py under test (run_ldm_sampler.py:42-45 sample layout).

A FRESH child job -- `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` with
LDM_DIST_BACKEND=gloo LDM_ONE_DEVICE=1, i.e. two ranks sharing the one GPU of the test box and
gathering over gloo -- runs the very code the driver launches at N > 1: rank r samples the global
indices [rB, (r+1)B), converts to uint8 per image, ONE all-gather.  Checked against two
single-process runs of the same script (`--first-sample-index 0` / `B`): the gathered bytes must
equal their concatenation byte for byte, and the JSON line must report world_size 2.

Children are started with subprocess (never exec from this process, which has initialised the GPU).
The model is bench.py's `--tiny` test model in float32 (same architecture, a few MB), 4 DDIM steps.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--tiny", "--dtype", "f32", "--decoder-dtype", "f32", "--latent", "16", "--ddim-steps", "4",
          "--batch-per-gpu", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-variant"]


def _run(cmd, env):
  r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
  assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
  assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}: {r.stdout[-1000:]}"
  return json.loads(lines[0])


def test_two_rank_bench_equals_single_process_runs(dev, tmp_path):
  bench = os.path.join(ROOT, "bench.py")
  base_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
  base_env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
  # two ranks, one device, gloo
  env2 = dict(base_env, LDM_DIST_BACKEND="gloo", LDM_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
  both = tmp_path / "both.npy"
  import socket
  with socket.socket() as sk:               # a free rendezvous port (a leftover child of an earlier run may hold a fixed one)
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
  j2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
             "--master-addr", "127.0.0.1", "--master-port", str(port), bench, "--gpus", "2",
             "--dump-images", str(both)] + COMMON, env2)
  assert j2["world_size"] == 2 and j2["n_gpus"] == 2 and j2["backend"] == "gloo"
  assert j2["scaling"] == "weak" and j2["config"]["global_batch"] == 4
  assert len(j2["ms_per_unet_step_per_rank"]) == 2
  got = np.load(both)
  assert got.dtype == np.uint8 and got.shape == (4, 128, 128, 3)
  # what each rank computes, as single processes
  halves = []
  for first in (0, 2):
    f = tmp_path / f"single{first}.npy"
    j1 = _run([sys.executable, bench, "--gpus", "1", "--first-sample-index", str(first),
               "--dump-images", str(f)] + COMMON, base_env)
    assert j1["world_size"] == 1 and j1["backend"] == "none"
    halves.append(np.load(f))
  ref = np.concatenate(halves, 0)
  assert np.array_equal(got, ref), f"{int((got != ref).sum())} bytes differ between the 2-rank and the single-process runs"
  # the two halves are different samples (x_T keyed by the global index), not one sample twice
  assert not np.array_equal(halves[0], halves[1])
