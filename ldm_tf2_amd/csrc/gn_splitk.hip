// ldm_groupnorm_splitk: a deferred split-K product completed INSIDE the GroupNorm that consumes it
// (gn_fused.h, SK = true).  Own translation unit: see the note in gn_fused.h.
#include "gn_fused.h"

using namespace ldm_gn;

// the split-K form keeps MAXCH * EPC float sums per thread: the plans with 24 chunks per thread or 1024
// threads would spill, they fall back to reduce + GroupNorm
static bool splitk_plan(int B, int HW, int C, int groups, int esize, GnFusedPlan* pl) {
  return gn_fused_plan(B, HW, C, groups, esize, pl) && pl->maxch <= 16 && pl->NT <= 512;
}

extern "C" int ldm_groupnorm_splitk_supported(int B, int HW, int C, int groups, int dtype) {
  if (!(dtype == LDM_F32 || dtype == LDM_BF16) || B <= 0 || HW <= 0 || C <= 0 || C % 4) return 0;
  GnFusedPlan pl;
  return splitk_plan(B, HW, C, groups, dtype == LDM_BF16 ? 2 : 4, &pl) ? 1 : 0;
}

extern "C" int ldm_groupnorm_splitk(const ldm_gemm_params* p, const float* gamma, const float* beta, void* gn_out,
                                    int64_t ld_gn, int B, int HW, int groups, float eps, int silu, int store_out,
                                    void* stream) {
  LDM_CHECK_ARG(p && gamma && beta && gn_out && p->workspace, "ldm_groupnorm_splitk: null pointer");
  LDM_CHECK_ARG(p->out_dtype == LDM_F32 || p->out_dtype == LDM_BF16, "ldm_groupnorm_splitk: bad dtype");
  const int split = ldm_gemm_splits(p);
  LDM_CHECK_ARG(split > 1, "ldm_groupnorm_splitk: these parameters do not split K (nothing was deferred)");
  LDM_CHECK_ARG(p->workspace_bytes >= (size_t)split * p->M * p->N * 4, "ldm_groupnorm_splitk: workspace too small");
  const int C = p->N;
  const int epc = p->out_dtype == LDM_BF16 ? 8 : 4;
  LDM_CHECK_ARG(B > 0 && HW > 0 && groups > 0 && C % groups == 0 && p->M == B * HW && p->batch == 1 &&
                    p->act == LDM_ACT_NONE && p->ldc_n == 1 && !p->out2 && !p->ln_out,
                "ldm_groupnorm_splitk: needs M == B*HW rows, a row-major output, no activation / out2 / ln_out");
  auto al16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
  LDM_CHECK_ARG(ld_gn % epc == 0 && al16(gn_out) && al16(p->workspace) && C % 4 == 0 &&
                    (!store_out || (p->out && p->ldc_m % epc == 0 && al16(p->out))) &&
                    (!p->residual || (p->ldr % epc == 0 && al16(p->residual))) &&
                    (!p->addend || (p->add_rows == HW)),
                "ldm_groupnorm_splitk: alignment (16-byte rows of out / gn_out / residual) or add_rows != HW");
  GnFusedPlan pl;
  LDM_CHECK_ARG(splitk_plan(B, HW, C, groups, 16 / epc, &pl),
                "ldm_groupnorm_splitk: shape B=%d HW=%d C=%d groups=%d not supported (ldm_groupnorm_splitk_supported; "
                "use ldm_gemm_reduce, then a plain GroupNorm)", B, HW, C, groups);
  GnSplitK sk;
  sk.ws = (const float*)p->workspace; sk.slab = (int64_t)p->M * p->N; sk.split = split; sk.N = p->N;
  sk.bias = p->bias; sk.addend = p->addend; sk.add_ld = p->add_ld; sk.residual = p->residual; sk.ldr = p->ldr;
  sk.xout = store_out ? p->out : nullptr; sk.alpha = p->alpha;
  dim3 grid(8 * (groups / pl.GB) * ((B + 7) / 8));
  hipStream_t s = (hipStream_t)stream;
  if (p->out_dtype == LDM_BF16)
    gn_fused_launch<bf16_t, true>(pl, grid, s, nullptr, p->ldc_m, gamma, beta, gn_out, ld_gn, B, HW, C, groups, eps, silu, sk);
  else
    gn_fused_launch<float, true>(pl, grid, s, nullptr, p->ldc_m, gamma, beta, gn_out, ld_gn, B, HW, C, groups, eps, silu, sk);
  return ldm_launch_status("ldm_groupnorm_splitk");
}

