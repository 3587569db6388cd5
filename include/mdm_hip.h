/* mdm_hip.h -- C ABI of libmdm_hip.so: the MI355X (gfx950) denoising hot path of
 * ltdoanh2004/MotionDiffusion-MoE.
 *
 * Conventions (every entry point):
 *   - raw DEVICE pointers + sizes, no torch types; `stream` is a hipStream_t passed as void*;
 *   - returns 0 on success (MDM_OK), non-zero status otherwise; never allocates, never synchronises,
 *     never copies to the host: safe to capture into a hipGraph;
 *   - tensors are fp32 row-major unless stated; "packed" weights are bf16 planes produced by
 *     mdm_pack_bf16 (hi plane, optional lo plane for the bf16x3 fp32-grade mode);
 *   - `precision`: 1 = single bf16 MFMA pass, 3 = bf16x3 split (hi*hi + hi*lo + lo*hi), fp32-grade.
 *
 * The reference has no FFI: its boundary is the Python call `model(x, t, **kwargs)`
 * (text2motion/models/gaussian_diffusion.py:493) resolved by MotionTransformer.forward
 * (text2motion/models/transformer.py:291-361).  Each entry point below names the reference lines it replaces;
 * the Python side that binds them (ctypes) is motiondiffusion-moe_amd/_lib.py, see INTEGRATION.md.
 */
#ifndef MDM_HIP_H
#define MDM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MDM_OK = 0, MDM_ERR_ARG = 1, MDM_ERR_LAUNCH = 2, MDM_ERR_UNSUPPORTED = 3 };
enum { MDM_OP_F32_ROW = 0, MDM_OP_F32_KSTRIDE = 1, MDM_OP_BF16_ROW = 2 };
enum { MDM_ACT_NONE = 0, MDM_ACT_GELU = 1, MDM_ACT_SILU = 2, MDM_ACT_FEAT = 3 };

/* One GEMM operand: a [rows x K] matrix seen through a loader kind (see csrc/gemm.h). */
typedef struct MdmOperand {
  const void* p;
  const void* p_lo;      /* BF16_ROW: lo plane (precision 3) or NULL */
  int64_t ld;            /* F32_ROW/BF16_ROW: elements between rows; F32_KSTRIDE: elements between k */
  int64_t gstride;       /* F32_ROW: row r -> (r / rpg) * gstride + (r % rpg) * ld when rpg > 0 */
  const int32_t* gather; /* F32_ROW: optional row gather */
  int64_t bs1, bs2;      /* batch strides: z -> (z / nb2) * bs1 + (z % nb2) * bs2 */
  int32_t rpg;
  int32_t kind;
} MdmOperand;

/* C = epilogue(A[M,K] * W[N,K]^T):
 *   v = alpha * (acc + bias[n]);  v = act(v);  v *= out_scale * colscale[n] * rowscale[m];
 *   v += r1_scale * R1[m (mod r1_mod), n] + R2[m, n]
 * Replaces every nn.Linear / einsum on the path (transformer.py:319-360, fast_attention.py:59-78,
 * 145-147,165,248-253,305-329, switch_moe.py:104, stylization.py:26-30). */
typedef struct MdmGemmDesc {
  MdmOperand A, W;
  int32_t M, N, K;
  int32_t batch, nb2;
  const int32_t* goff; /* grouped mode: row ranges [goff[g], goff[g+1]) use W + g*W.bs1, bias + g*bias_bs */
  int32_t ngroups;
  int32_t act;
  float* C;
  int64_t ldc, c_bs1, c_bs2;
  const float* bias;
  int64_t bias_bs;
  float alpha, out_scale, r1_scale;
  int32_t r1_mod;
  const float* colscale;
  const float* rowscale;
  const float* R1;
  int64_t ldr1;
  const float* R2;
  int64_t ldr2;
  const int32_t* feat_len; /* ACT_FEAT key masking (fast_attention.py:69-74) */
  int32_t feat_S, feat_rpt, feat_kslot;
  int32_t precision;
} MdmGemmDesc;

int mdm_gemm(const MdmGemmDesc* desc, void* stream);

/* fp32 [rows, K] (row stride ld_src) -> bf16 planes [rows, Kpad] (Kpad = ld_dst, multiple of 32, zero padded);
 * lo may be NULL.  Weight packing happens once at load time (not on the hot path). */
int mdm_pack_bf16(const float* src, int64_t ld_src, int64_t rows, int64_t K, uint16_t* hi, uint16_t* lo,
                  int64_t ld_dst, void* stream);

const char* mdm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MDM_HIP_H */
