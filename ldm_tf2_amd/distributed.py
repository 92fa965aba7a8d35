"""Multi-GPU sampling: independent latent samples sharded over ranks, one all-gather.

The sampling path shards by construction: every sample's
trajectory depends only on its own x_T, its context row and the (replicated)
weights.  One process per GPU, launched by torch.distributed.run; rank r owns the
global sample indices [r*B, (r+1)*B); x_T is keyed by the global sample index so
an N-GPU run reproduces the 1-GPU run sample for sample.  The only communication
is ONE all-gather of the decoded images after the loop (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).  No collective exists inside the loop.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
  """Returns (rank, world_size, local_rank).  Single process when WORLD_SIZE is unset."""
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  if world > 1 and not dist.is_initialized():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
      # LDM_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks
      backend = os.environ.get("LDM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if os.environ.get("LDM_ONE_DEVICE"):
      local = 0                               # rehearsal: every rank on device 0
    if backend == "nccl":
      torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
  return rank, world, (0 if os.environ.get("LDM_ONE_DEVICE") else local)


def shard_range(rank, batch_per_rank):
  """Global sample indices owned by `rank`."""
  first = rank * batch_per_rank
  return first, first + batch_per_rank


def all_gather_images(images):
  """[B,H,W,3] per rank -> [world*B,H,W,3] on every rank, in rank order: the one
  collective of the path.  Identity when not distributed."""
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    return images
  images = images.contiguous()
  out = torch.empty((dist.get_world_size() * images.shape[0],) + tuple(images.shape[1:]),
                    dtype=images.dtype, device=images.device)
  if images.is_cuda and dist.get_backend() == "gloo":   # rehearsal only: gloo gathers host tensors
    host = torch.empty(out.shape, dtype=out.dtype)
    dist.all_gather_into_tensor(host, images.cpu())
    out.copy_(host)
    return out
  dist.all_gather_into_tensor(out, images)
  return out


def barrier():
  if dist.is_available() and dist.is_initialized():
    dist.barrier()


def max_over_ranks(value: float, device) -> float:
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    return float(value)
  if dist.get_backend() == "gloo":
    device = "cpu"
  t = torch.tensor([float(value)], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())


def world_size() -> int:
  """Rank count as torch.distributed reports it (1 when not initialised)."""
  return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def backend_name() -> str:
  """'nccl' is RCCL on ROCm; 'none' for a single process."""
  return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else "none"


def gather_floats(value: float, device):
  """One float per rank, in rank order, on every rank (bench.py: per-rank loop times)."""
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    return [float(value)]
  if dist.get_backend() == "gloo":
    device = "cpu"
  t = torch.tensor([float(value)], dtype=torch.float64, device=device)
  out = torch.empty(dist.get_world_size(), dtype=torch.float64, device=device)
  dist.all_gather_into_tensor(out, t)
  return [float(v) for v in out.cpu()]
