"""ctypes binding of libldm_hip.so (the C ABI declared in include/ldm_hip.h).

The product path has NO fallback: if the shared library is missing or does not
export every symbol of the header, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LDM_HIP_LIB=<file>: load another build of the same ABI (tools/ use it for the -DLDM_TOOLS_BUILD
# library with its timing ablations, `make tools`); the product library itself reads no
# result-changing environment variable
LIB_PATH = os.environ.get("LDM_HIP_LIB") or os.path.join(_HERE, "lib", "libldm_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_GEGLU, ACT_SILU = 0, 1, 2, 3
OK, ERR_ARG, ERR_LAUNCH, ERR_WORKSPACE = 0, -1, -2, -3     # include/ldm_hip.h status codes

c_i64, c_i32, c_f32, c_vp, c_sz = C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_size_t


class GemmParams(C.Structure):
  """Mirror of `ldm_gemm_params` (include/ldm_hip.h)."""
  _fields_ = [
      ("a", c_vp), ("w", c_vp), ("bias", c_vp), ("addend", c_vp), ("residual", c_vp),
      ("out", c_vp), ("workspace", c_vp), ("workspace_bytes", c_sz),
      ("lda", c_i64), ("ldr", c_i64), ("ldc_m", c_i64), ("ldc_n", c_i64),
      ("stride_a", c_i64), ("stride_w", c_i64), ("stride_c", c_i64), ("stride_r", c_i64),
      ("add_ld", c_i64),
      ("M", c_i32), ("N", c_i32), ("K", c_i32), ("batch", c_i32),
      ("add_rows", c_i32),
      ("conv", c_i32), ("B", c_i32), ("H", c_i32), ("W", c_i32), ("Cin", c_i32),
      ("OH", c_i32), ("OW", c_i32), ("stride", c_i32), ("upsample", c_i32),
      ("act", c_i32), ("dtype", c_i32), ("out_dtype", c_i32), ("split_k", c_i32),
      ("tile", c_i32), ("alpha", c_f32),
      ("no_lead_pad", c_i32),
      ("ln_out", c_vp), ("ln_gamma", c_vp), ("ln_beta", c_vp), ("ld_ln", c_i64), ("ln_eps", c_f32),
      ("out2", c_vp), ("ld2", c_i64), ("stride2", c_i64), ("n_split", c_i32), ("rows2", c_i32),
      ("ln_cs", c_vp), ("defer_reduce", c_i32),
      ("a2", c_vp), ("lda2", c_i64), ("Cin2", c_i32),
  ]


# name -> (restype, argtypes); must list EVERY function of include/ldm_hip.h
SIGNATURES = {
    "ldm_version": (c_i32, []),
    "ldm_last_error": (c_i32, [C.c_char_p, c_i32]),
    "ldm_gemm": (c_i32, [C.POINTER(GemmParams), c_vp]),
    "ldm_gemm_workspace_bytes": (c_sz, [C.POINTER(GemmParams)]),
    "ldm_gemm_splits": (c_i32, [C.POINTER(GemmParams)]),
    "ldm_gemm_reduce": (c_i32, [C.POINTER(GemmParams), c_vp]),
    "ldm_groupnorm_splitk_supported": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "ldm_groupnorm_splitk": (c_i32, [C.POINTER(GemmParams), c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_f32,
                                     c_i32, c_i32, c_vp]),
    "ldm_conv3x3_small": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_i32,
                                  c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "ldm_groupnorm_nchunks": (c_i32, [c_i32, c_i32, c_i32]),
    "ldm_groupnorm_partial": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32,
                                      c_i32, c_vp]),
    "ldm_groupnorm_apply": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32,
                                    c_i32, c_i32, c_i32, c_f32, c_i32, c_i32, c_vp]),
    "ldm_layernorm": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_f32,
                              c_i32, c_vp]),
    "ldm_softmax_rows": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_f32,
                                 c_vp]),
    "ldm_attention": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64,
                              c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32,
                              c_i32, c_vp]),
    "ldm_attention_ms": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64,
                                 c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "ldm_st_tail": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                            c_vp, c_i64, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "ldm_st_xtail": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp,
                             c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "ldm_st_block": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32,
                             c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32,
                             c_f32, c_i32, c_vp]),
    "ldm_ffn_geglu_supported": (c_i32, [c_i32, c_i32, c_i32]),
    "ldm_ffn_geglu": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "ldm_time_embedding": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "ldm_select_row": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "ldm_gemv": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32,
                         c_i32, c_i32, c_vp]),
    "ldm_cfg_ddim_update": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32,
                                    c_f32, c_i32, c_i32, c_i64, c_vp]),
    "ldm_post_quant": (c_i32, [c_vp, c_f32, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "ldm_groupnorm_fused_supported": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "ldm_groupnorm_fused": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32,
                                    c_i32, c_i32, c_vp]),
    "ldm_gemm_plan": (c_i32, [c_vp, c_vp, c_vp]),
    "ldm_gemm_ln_supported": (c_i32, [c_i32, c_i32]),
    "ldm_gaussian_sample": (c_i32, [c_vp, c_vp, c_vp, c_f32, c_i64, c_i32, c_vp]),
    "ldm_vq_nearest": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "ldm_embedding": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "ldm_minmax_u8": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i32, c_i64, c_vp]),
    "ldm_cast": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_i64, c_i32, c_vp]),
}


class LdmHipError(RuntimeError):
  pass


def _load():
  if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make` (or `python -c 'import __graft_entry__ as g; "
        "g.build()'`).  There is no CPU fallback for the sampling path.")
  lib = C.CDLL(LIB_PATH)
  for name, (res, args) in SIGNATURES.items():
    try:
      fn = getattr(lib, name)
    except AttributeError as e:
      raise ImportError(f"{LIB_PATH} does not export {name}") from e
    fn.restype = res
    fn.argtypes = args
  return lib


lib = _load()


def last_error() -> str:
  buf = C.create_string_buffer(512)
  lib.ldm_last_error(buf, 512)
  return buf.value.decode(errors="replace")


def check(status: int, what: str = ""):
  if status != 0:
    raise LdmHipError(f"{what or 'ldm_hip'} failed ({status}): {last_error()}")
