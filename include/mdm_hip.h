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
enum { MDM_OP_F32_ROW = 0, MDM_OP_F32_KSTRIDE = 1, MDM_OP_BF16_ROW = 2, MDM_OP_FP8_ROW = 3 /* e4m3 bytes, csrc/gemm8.hip */,
       /* activation rows PRE-SPLIT for the fp32-grade kernel (csrc/gemm3.hip): per row and per block of 32 k, 32 bf16 "hi" then 32
        * bf16 "lo" = rn(x - hi) -- 128 bytes, exactly the bytes of the 32 fp32 values they replace, so `ld` (in 4-byte units, as for
        * MDM_OP_F32_ROW) and every buffer size stay what they are for fp32 rows.  Written by the producers (MdmGemmDesc.Cx2 and the
        * row-wise / attention kernels) so that the GEMM's K loop does not re-split each fp32 fragment in every wave that reads it */
       MDM_OP_X2_ROW = 4 };
enum { MDM_ACT_NONE = 0, MDM_ACT_GELU = 1, MDM_ACT_SILU = 2, MDM_ACT_FEAT = 3,
       /* fp32-grade kernel only (precision 3, N % 128 == 0): every 128-column slice of a row is one attention head --
        * LayerNorm over the slice with hn_w / hn_b (eps 1e-5), L2-normalised when the slice index is < hn_l2_tiles
        * (fast_attention.py:44-55 on the q | k | v projection), written as bf16 hi / lo planes C16 / C16_lo */
       MDM_ACT_HEADNORM = 4,
       /* the same kernel: softmax over every 128-column slice (the head_dim softmax of the linear cross-attention's query,
        * fast_attention.py:248), written as bf16 hi / lo planes C16 / C16_lo */
       MDM_ACT_HEADSOFTMAX = 5 };
/* 16-bit operand / storage format of the single-pass MFMA kernels: bf16, or IEEE fp16 (same MFMA rate on gfx950, 8x finer
 * rounding, |x| <= 65504: used where the value range is known). */
enum { MDM_H16_BF16 = 1, MDM_H16_F16 = 2 };
/* `precision` of the model-level entry points:
 *   1  single bf16 MFMA pass, GEMM-only tensors stored bf16 (throughput; BASELINE configs[1] names bf16)
 *   2  single fp16 MFMA pass, GEMM-only tensors stored fp16 (same speed, ~8x smaller error)
 *   3  bf16x3 split products everywhere, fp32 activations (fp32-grade: the mode that meets the 1e-3 parity bar)
 *   4  mixed: bf16x3 for everything that feeds the fp32 residual stream and the MoE router, single fp16 pass for the
 *      MFMA-bound GEMMs only (expert MLPs, the 4x FFN of the text cross-attention block)
 *   5  as 2, with the expert GEMMs on fp8 (e4m3) operands: activations quantised per row by the router kernel, weights per
 *      output channel at pack time, block-scaled MFMA with unit block scales (csrc/gemm8.hip); BASELINE configs[4]
 * The packed weights must be in the matching format (packing.py: weight_format). */
enum { MDM_PREC_BF16 = 1, MDM_PREC_F16 = 2, MDM_PREC_X3 = 3, MDM_PREC_MIXED = 4, MDM_PREC_FP8 = 5 };

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
  float* C;       /* fp32 output (may be NULL when C16 is set) */
  int64_t ldc, c_bs1, c_bs2;
  uint16_t* C16;  /* optional bf16 copy of the output (same ldc / batch strides, in elements) */
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
  int32_t precision; /* 1 single pass, 3 bf16x3; 2 = single pass with fp16 operands (sets h16) */
  int32_t h16;       /* MDM_H16_*: format of BF16_ROW activation / weight planes in the single-pass kernels and of C16 */
  /* fp8 GEMM (A.kind == W.kind == MDM_OP_FP8_ROW): acc * a_scale_u * a_scale[src row of m] * w_scale[n] (+ bias, act, ...) */
  const float* a_scale; /* per activation row (indexed by the GATHERED source row), or NULL */
  const float* w_scale; /* per output channel; grouped like bias (bias_bs), or NULL */
  float a_scale_u;      /* uniform activation scale (1 by default) */
  uint8_t* C8;          /* optional fp8 output e4m3(v * c8_scale), same ldc */
  float c8_scale;
  /* weight-gradient mode (both operands MDM_OP_F32_KSTRIDE, no goff): batch z reduces over the K range
   * [kgoff[z], kgoff[z+1]) -- the routed rows of expert group z -- instead of [0, K); an empty range writes epilogue(0) */
  const int32_t* kgoff;
  /* MDM_ACT_HEADNORM: LayerNorm weight / bias over head_dim = 128, number of leading 128-column slices that are also
   * L2-normalised (2 H for q | k | v), and the lo plane of the output (C16 = hi plane; hi + lo = the fp32 value to ~2^-17).
   * C16_lo with MDM_ACT_NONE / HEADSOFTMAX on the fp32-grade kernel: the result leaves as bf16 hi / lo planes (C must be NULL) */
  const float* hn_w;
  const float* hn_b;
  uint16_t* C16_lo;
  int32_t hn_l2_tiles;
  /* fp32-grade kernel: the result as MDM_OP_X2_ROW rows (row stride 2 * ldc 16-bit elements; N % 32 == 0), beside or instead of C */
  uint16_t* Cx2;
  /* optional fragment stream of W (mdm_gemm_stream1_pack, format h16): a plain Linear on 16-bit rows (precision 1 / 2, no batch /
   * groups / gather, act NONE or GELU, N % 256 == K % 256 == 0) then runs on the streamed-weight kernel (csrc/gemm_stream.hip);
   * anything else ignores it */
  const uint16_t* w_stream;
  /* fp32-grade mode (precision 3) with MDM_OP_X2_ROW activations: w_stream = the (bf16 hi, lo) fragment-pair stream of W
   * (mdm_gemm_stream3x_pack; K in {512, 1024}, N % 512 == 0) selects the streamed-weight bf16x3 kernel (csrc/gemm_stream3.hip);
   * grouped launches (goff) give the stream's elements per group (mdm_gemm_stream3x_elems(1, N, K) minus the tail pad) here */
  int64_t w_stream_gs;
} MdmGemmDesc;

int mdm_gemm(const MdmGemmDesc* desc, void* stream);

/* Fused two-layer MLP (throughput mode, bf16 operands, fp32 accumulation):
 *   Y[m,:] = ( GELU(X[src(m),:] W1^T + b1) W2^T + b2 ) * rowscale[m] + r1_scale * R1[m,:] + R2[m,:]
 * the hidden activations stay on chip.  Replaces the Linear-GELU-Linear pairs of the path: the expert MLPs
 * (switch_moe.py:19-25, grouped by goff with per-expert strides w?_gs / b?_gs), the 4x FFN of the text cross-attention
 * block (fast_attention.py:293-299) and the Performer output projection (fast_attention.py:121-126).
 * Supported shapes: Dout == 512, Din % 64 == 0, F % 256 == 0 (LDS-staged kernel); with a weight stream (below) Dout == 512 and
 * Din in {128, 256, 512}, or Dout == Din == 1024; anything else returns MDM_ERR_UNSUPPORTED. */
typedef struct MdmMlpDesc {
  const uint16_t* X; /* bf16 rows [*, Din] */
  int64_t ldx;
  const int32_t* gather; /* optional: row m reads X[gather[m]] */
  int32_t M, Din, F, Dout;
  const int32_t* goff;
  int32_t ngroups;
  const uint16_t* w1; /* bf16 [F, Din] (row stride ldw1) */
  int64_t ldw1, w1_gs;
  const float* b1;
  int64_t b1_gs;
  const uint16_t* w2; /* bf16 [Dout, F] (row stride ldw2) */
  int64_t ldw2, w2_gs;
  const float* b2;
  int64_t b2_gs;
  const float* rowscale;
  const float* R1;
  int64_t ldr1;
  float r1_scale;
  const float* R2;
  int64_t ldr2;
  float* C;      /* fp32 output (may be NULL when C16 is set) */
  uint16_t* C16; /* optional 16-bit copy */
  int64_t ldc;
  int32_t h16;   /* MDM_H16_*: format of X, w1, w2 and C16 (0 = bf16) */
  /* optional weight STREAM built by mdm_mlp_stream_pack from the same w1 / w2 (same 16-bit format): per (group, wave) one
   * linear run of 1-KiB MFMA fragments in consumption order (csrc/mlp_stream.hip).  When set and the shape is one of the
   * streamed kernel's (above; F % 256 == 0, ngroups <= 64) that kernel runs and w1 / w2 are not read; wstream_gs = elements
   * per group = F * Din + Dout * F.  The buffer must have mdm_mlp_stream_elems() elements (16 KiB of tail padding). */
  const uint16_t* wstream;
  int64_t wstream_gs;
} MdmMlpDesc;

int mdm_fused_mlp(const MdmMlpDesc* desc, void* stream);
/* Weight stream of the fused MLP: elements the buffer needs, and the packer (fp32 row-major w1 [G, F, Din], w2 [G, Dout, F]
 * -> h16-format stream; once at load time). */
int64_t mdm_mlp_stream_elems(int32_t G, int32_t F, int32_t Din, int32_t Dout);
int mdm_mlp_stream_pack(const float* w1, const float* w2, int32_t G, int32_t F, int32_t Din, int32_t Dout, int32_t h16,
                        uint16_t* out, void* stream);
/* Weight stream of ONE Linear [N, K] fp32 for the fused stylization kernel (csrc/style_gemm.hip; N = K = 512): elements
 * the buffer needs (0 = shape not taken) and the packer. */
int64_t mdm_gemm_stream_elems(int32_t N, int32_t K);
int mdm_gemm_stream_pack(const float* w, int32_t N, int32_t K, int32_t h16, uint16_t* out, void* stream);
/* Fragment stream of a [N, K] fp32 weight for the streamed-weight GEMM (MdmGemmDesc.w_stream / MdmPacked.ws): [N / 16][K / 32]
 * MFMA operand fragments of 1 KiB in the 16-bit format h16, plus the read-ahead pad.  elems: 16-bit elements to allocate, 0 when the
 * shape is not covered (N % 256, K % 256). */
int64_t mdm_gemm_stream1_elems(int32_t N, int32_t K);
/* ... and of G stacked [N, K] fp32 weights (the experts of one layer) as (bf16 hi, lo) fragment PAIRS for the streamed-weight
 * bf16x3 GEMM (MdmGemmDesc.w_stream with MDM_OP_X2_ROW activations); group_elems = the stride between groups (w_stream_gs) */
int64_t mdm_gemm_stream3x_elems(int32_t G, int32_t N, int32_t K);
int64_t mdm_gemm_stream3x_group_elems(int32_t N, int32_t K);
int mdm_gemm_stream3x_pack(const float* w, int64_t ldw, int32_t G, int32_t N, int32_t K, uint16_t* out, void* stream);
int mdm_gemm_stream1_pack(const float* w, int64_t ldw, int32_t N, int32_t K, int32_t h16, uint16_t* out, void* stream);

/* the same for the fp32-grade (bf16x3) form of that kernel: (bf16 hi, lo = rn(w - hi)) fragment PAIRS in consumption order,
 * 2 * N * K + 16 KiB of tail padding elements (0 = shape not taken) */
int64_t mdm_gemm_stream3_elems(int32_t N, int32_t K);
int mdm_gemm_stream3_pack(const float* w, int32_t N, int32_t K, uint16_t* out, void* stream);

/* fp32 [rows, K] (row stride ld_src) -> bf16 planes [rows, Kpad] (Kpad = ld_dst, multiple of 32, zero padded);
 * lo may be NULL.  Weight packing happens once at load time (not on the hot path). */
int mdm_pack_bf16(const float* src, int64_t ld_src, int64_t rows, int64_t K, uint16_t* hi, uint16_t* lo,
                  int64_t ld_dst, void* stream);
/* fp32 [rows, K] -> e4m3 bytes [rows, ld_dst] (ld_dst multiple of 128, zero padded) + scales[rows] = amax / 448 */
int mdm_pack_fp8(const float* src, int64_t ld_src, int64_t rows, int64_t K, uint8_t* dst, int64_t ld_dst, float* scales,
                 void* stream);
/* same layout, one IEEE fp16 plane (MDM_H16_F16 weights of the single-pass kernels) */
int mdm_pack_f16(const float* src, int64_t ld_src, int64_t rows, int64_t K, uint16_t* dst, int64_t ld_dst, void* stream);

/* ---- packed weights of one denoiser (built once at load time by motiondiffusion-moe_amd/packing.py) ------------ */
typedef struct MdmPacked { /* 16-bit planes of an fp32 [N,K] weight, K padded to ld (multiple of 32) */
  const uint16_t* hi; /* bf16 hi plane, or the fp16 plane of a weight packed for a single fp16 pass */
  const uint16_t* lo; /* bf16 lo plane (bf16x3); for an fp8-packed weight (hi = e4m3 bytes): its per-row fp32 scales; else NULL */
  int64_t ld;
  const uint16_t* ws; /* optional fragment stream of the same weight in the model's 16-bit format (mdm_gemm_stream1_pack), or NULL */
} MdmPacked;

typedef struct MdmStyle { /* StylizationBlock minus its emb_layers (those are stacked model-wide), stylization.py:5-31 */
  const float *norm_w, *norm_b;
  MdmPacked out; /* out_layers.2 [D,D] */
  const float* out_b;
  const uint16_t* out_ws; /* optional weight stream of out_layers.2 (mdm_gemm_stream_pack; 16-bit modes, D == 512), or NULL */
  const uint16_t* out_ws3; /* optional (hi, lo) pair stream of out_layers.2 (mdm_gemm_stream3_pack; fp32-grade modes, D == 512), or NULL */
} MdmStyle;

typedef struct MdmPerformer { /* PerformerSelfAttention, fast_attention.py:94-179 */
  const float *pre_w, *pre_b, *post_w, *post_b;
  MdmPacked qkv; /* query|key|value stacked [3D, D] (fast_attention.py:145-147) */
  const float* qkv_b;
  const float *hn_w, *hn_b; /* fast_attention.norm over head_dim */
  MdmPacked feat;           /* projection_matrix^T [m, dh] (captured random state, fast_attention.py:19-36) */
  MdmPacked proj0, proj3;
  const float *proj0_b, *proj3_b;
  const uint16_t* proj_ws; /* optional weight stream of proj_out.0 / proj_out.3 (mdm_mlp_stream_pack, G = 1), or NULL */
  MdmStyle style;
} MdmPerformer;

typedef struct MdmLayer { /* MoEExtendedDecoderLayer, transformer.py:17-64 */
  const float *dual_pre_w, *dual_pre_b, *dual_post_w, *dual_post_b;
  MdmPerformer local, global;
  MdmPacked skip;
  const float* skip_b;
  /* GatedCrossAttention / LinearTemporalCrossAttention, fast_attention.py:227-272 */
  const float *ca_norm_w, *ca_norm_b, *ca_tnorm_w, *ca_tnorm_b;
  MdmPacked ca_q, ca_k, ca_v;
  const float *ca_q_b, *ca_k_b, *ca_v_b;
  const float* ca_gvec; /* sigmoid(gate) * sigmoid(adaptive_gate), [D] */
  MdmStyle ca_style;
  /* MoEMultiBranchFFN / SwitchMoELayer, multi_branch.py:31-61, switch_moe.py:7-111; experts stacked branch-major */
  const float *moe_ln_w[2], *moe_ln_b[2], *gate_w[2], *gate_b[2];
  MdmPacked w1; /* [2*E*F, D] */
  MdmPacked w2; /* [2*E*D, F] */
  const uint16_t* wstream;   /* optional weight stream of the 2E expert MLPs (mdm_mlp_stream_pack; 16-bit expert modes), or NULL */
  int64_t wstream_gs;        /* elements per expert group: F * D + D * F */
  const float *b1, *b2;
  float *usage[2], *importance[2]; /* expert_usage / expert_importance buffers, updated in place; may be NULL */
  MdmStyle ffn_style;
  /* MemoryEfficientCrossAttentionBlock, fast_attention.py:274-330 */
  MdmPacked sd_q, sd_k, sd_v, sd_out, sd_f1, sd_f2;
  const float *sd_q_b, *sd_k_b, *sd_v_b, *sd_out_b, *sd_ln_w, *sd_ln_b, *sd_f1_b, *sd_f2_b;
  /* optional fp32 copies of sd_cross_attn.query / .out weights [D, D]: needed only to build the folded text cache */
  const float *sd_q_w32, *sd_out_w32;
  const uint16_t* sd_ffn_ws; /* optional weight stream of sd_cross_attn.ffn.1 / .3 (mdm_mlp_stream_pack, G = 1), or NULL */
} MdmLayer;

typedef struct MdmModel { /* MotionTransformer, transformer.py:166-361 */
  int32_t D, F, Dt, H, E, L, feats, num_frames;
  MdmPacked tmlp0, tmlp2, te0, te2, tproj, gf_time, gf_text, gf_post0, gf_post2, text_proj, joint, down, up, out;
  const float *tmlp0_b, *tmlp2_b, *te0_b, *te2_b, *tproj_b, *gf_time_b, *gf_text_b, *gf_post0_b, *gf_post2_b,
      *text_proj_b, *joint_b, *down_b, *up_b2, *out_b;
  const float* seq_emb;  /* [num_frames, D] */
  MdmPacked style_eph;   /* the 8L per-call random emb projections stacked [8L*Te, D] (stylization.py:22-24) */
  const float* style_eph_b;
  MdmPacked style_emb;   /* emb_layers.1 of the 8L StylizationBlocks stacked [8L*2D, Te] */
  const float* style_emb_b;
  const MdmLayer* layers; /* 2L entries: low blocks then high blocks; style slot of layer i = 4*i + {local,global,cross,ffn} */
} MdmModel;

/* x/t-independent text-side state, one slab per decoder layer (2L of them) */
typedef struct MdmTextCache {
  float* lin_at; /* [2L, B, H, dh, dh]  A^T of fast_attention.py:252 */
  float* sd_k;   /* [2L, B, N, D]       key(xf)   of fast_attention.py:306 */
  float* sd_v;   /* [2L, B, N, D]       value(xf) of fast_attention.py:307 */
  int32_t B, N;
  /* optional (all three or none; throughput mode, D == 512, H * N <= 128): the query / output projections of the text
   * cross-attention folded into the text side, see csrc/sdfold.hip.  Zero-initialised by the caller (padding). */
  uint16_t* sd_kfold; /* 16-bit [2L, B, P, 128, D]  K'[hs*N + n, :] = key_h[n, :] Wq_h / sqrt(dh); P = mdm_sd_fold_passes, hs = h mod heads-per-pass */
  float* sd_cb;       /* fp32   [2L, B, P, 128]     key_h[n, :] . bq_h / sqrt(dh) */
  uint16_t* sd_vfold; /* 16-bit [2L, B, P, D, 128]  V'^T[:, hs*N + n] = Wout[:, h] value_h[n, :]^T */
  /* optional int32 [B] on the device, 1 <= ntok[b] <= N: sample b's own text token count; rows ntok[b] .. N-1 of its xf_out
   * are padding that neither cross-attention sees (weight exactly 0 in both softmaxes).  NULL = every sample has N tokens.
   * The reference has no text mask (fast_attention.py:249,317-320): this exists so that forwards the reference runs
   * separately because their captions tokenise to different lengths -- the cond and uncond halves of a guided step,
   * gaussian_diffusion.py:1060-1073 -- can travel as rows of one batch and still see exactly their own tokens. */
  const int32_t* ntok;
} MdmTextCache;

/* Optional per-loop stem cache: the time-embedding chain (time.py:15-31 -> time_embed -> time_proj -> gated_fusion.proj_time,
 * transformer.py:318-320, gate.py:16) depends only on the integer timestep, and gated_fusion.proj_text(text_proj(xf_proj))
 * only on the text: both are tabulated once per sampling loop instead of being recomputed by 8 tiny GEMMs every step. */
typedef struct MdmStemCache {
  const float* time_table; /* [steps, D], row t = proj_time(...(t)) */
  const float* gx;         /* [B, D] = proj_text(text_proj(xf_proj)) */
  int32_t steps;
} MdmStemCache;

/* Fill time_table [steps, D] (t = 0..steps-1) and gx [B, D]; either output may be NULL.  ws as for (B=128, T=2, N=1). */
int mdm_stem_cache_build(const MdmModel* m, int32_t steps, float* time_table, const float* xf_proj, int32_t B, float* gx,
                         void* ws, int64_t ws_bytes, int32_t precision, void* stream);

/* Bytes of scratch mdm_denoiser_forward / the block entry points need for (B, T, N). */
int64_t mdm_workspace_bytes(const MdmModel* m, int32_t B, int32_t T, int32_t N);

/* Build the text cache from xf_out [B, N, Dt] (hoists fast_attention.py:249-252,306-307 out of the step loop). */
int mdm_text_cache_build(const MdmModel* m, const float* xf_out, const MdmTextCache* tc, void* ws, int64_t ws_bytes,
                         int32_t precision, void* stream);

/* MotionTransformer.forward (transformer.py:291-361) with text already encoded:
 * x [B,T,feats], timesteps int64 [B], length int32 [B], xf_proj [B,Dt] -> out [B,T,feats].
 * forced_routing: NULL, or int32 [2L][2][B*S_layer][2] expert indices (parity tests); trace: NULL or per-block dumps;
 * stem: NULL or a cache built by mdm_stem_cache_build for integer timesteps in [0, steps). */
int mdm_denoiser_forward(const MdmModel* m, const MdmTextCache* tc, const float* x, const int64_t* timesteps,
                         const int32_t* length, const float* xf_proj, int32_t B, int32_t T, float* out, void* ws,
                         int64_t ws_bytes, const int32_t* forced_routing, float* trace, const MdmStemCache* stem,
                         int32_t precision, void* stream);

/* One decoder layer / its four blocks on h [B,S,D] in place semantics (out may alias nothing);
 * sc = the layer's 4 style (scale|shift) rows [4, B, 2D]; len int32 [B] (already halved for the low scale). */
enum { MDM_BLOCK_DUAL = 0, MDM_BLOCK_CROSS = 1, MDM_BLOCK_MOE = 2, MDM_BLOCK_SDCROSS = 3, MDM_BLOCK_LAYER = 4 };
int mdm_block_forward(const MdmModel* m, int32_t layer, int32_t block, const MdmTextCache* tc, const float* h,
                      const float* sc, const int32_t* len, int32_t B, int32_t S, float* out, void* ws, int64_t ws_bytes,
                      const int32_t* forced_routing, int32_t precision, void* stream);

/* The same blocks under the names SURVEY.md 8(b) lists (thin views of mdm_block_forward; sc4 = the layer's four
 * (scale|shift) rows [4, B, 2D] as above), plus one PerformerSelfAttention alone (fast_attention.py:137-179; which = 0
 * local_attn, 1 global_attn; sc = that block's (scale|shift) rows [B, 2D]; out = x + 0.1 * style(...)). */
int mdm_moe_ffn_forward(const MdmModel* m, int32_t layer, const float* h, const float* sc4, const int32_t* len, int32_t B,
                        int32_t S, float* out, void* ws, int64_t ws_bytes, const int32_t* forced_routing,
                        int32_t precision, void* stream);
int mdm_dual_self_attn_forward(const MdmModel* m, int32_t layer, const float* h, const float* sc4, const int32_t* len,
                               int32_t B, int32_t S, float* out, void* ws, int64_t ws_bytes, int32_t precision,
                               void* stream);
int mdm_linear_xattn_forward(const MdmModel* m, int32_t layer, const MdmTextCache* tc, const float* h, const float* sc4,
                             const int32_t* len, int32_t B, int32_t S, float* out, void* ws, int64_t ws_bytes,
                             int32_t precision, void* stream);
int mdm_softmax_xattn_ffn_forward(const MdmModel* m, int32_t layer, const MdmTextCache* tc, const float* h,
                                  const float* sc4, const int32_t* len, int32_t B, int32_t S, float* out, void* ws,
                                  int64_t ws_bytes, int32_t precision, void* stream);
int mdm_performer_attn_forward(const MdmModel* m, int32_t layer, int32_t which, const float* h, const float* sc,
                               const int32_t* len, int32_t B, int32_t S, float* out, void* ws, int64_t ws_bytes,
                               int32_t precision, void* stream);

/* StylizationBlock.forward given (scale|shift) = emb_layers(emb): out = Lin(SiLU(LN(h)*(1+scale)+shift)) */
int mdm_stylization_forward(const MdmStyle* st, const float* h, const float* sc, int32_t B, int32_t S, int32_t D,
                            float* tmp, float* out, int32_t precision, void* stream);

/* Stem: fused time/text embedding (transformer.py:313-321) and all 8L stylization (scale|shift) rows.
 * emb_out [B,D] (may be NULL), sc_out [8L, B, 2D]. */
int mdm_stem_embeddings(const MdmModel* m, const int64_t* timesteps, const float* xf_proj, int32_t B, float* emb_out,
                        float* sc_out, void* ws, int64_t ws_bytes, int32_t precision, void* stream);

/* Sampler updates on n = B*T*feats elements.  tab = fp32 schedule table [7][steps] with rows
 * sqrt_recip_acp, sqrt_recipm1_acp, coef1, coef2, post_logvar_clipped, acp, acp_prev
 * (gaussian_diffusion.py:405-431); t from *t_dev when non-NULL else t_imm.
 * CFG DDPM step (gaussian_diffusion.py:1042-1098; eps_u NULL = unguided p_sample :582-614; noise NULL = no noise;
 * clip_denoised clamps each pred_xstart to [-1,1] before guidance, :523-528). */
int mdm_cfg_posterior_step(const float* x, const float* eps_c, const float* eps_u, const float* noise, int64_t n,
                           const float* tab, int32_t steps, const int32_t* t_dev, int32_t t_imm, float cfg_scale,
                           int32_t clip_denoised, float* x_out, float* x0_out, void* stream);
/* DDIM step (gaussian_diffusion.py:699-742). */
int mdm_ddim_step(const float* x, const float* eps, const float* noise, int64_t n, const float* tab, int32_t steps,
                  const int32_t* t_dev, int32_t t_imm, float eta, int32_t clip_denoised, float* x_out, float* x0_out,
                  void* stream);

/* Counter-based gaussian noise (Philox4x32-10 + Box-Muller, csrc/noise.hip): out[s, e] for s < nsamples, e < per_sample is
 * a function of (seed, sample0 + s, stream, e) only, where stream = *stream_dev when non-NULL (the device-resident timestep
 * of a captured step) else stream_imm (MDM_NOISE_STREAM_XT for the initial x_T).  Replaces th.randn(*shape) /
 * th.randn_like(x) of gaussian_diffusion.py:1119,1094 where results must not depend on how the batch is sharded. */
enum { MDM_NOISE_STREAM_XT = 0x7fffffff };
int mdm_noise_normal(float* out, int64_t per_sample, int32_t nsamples, int64_t sample0, uint64_t seed,
                     const int32_t* stream_dev, int32_t stream_imm, void* stream);
/* the same with one explicit global sample index per row (device int64 [nsamples]): length-bucketed batches whose rows are
 * not consecutive samples (trainer.generate_bucketed) */
int mdm_noise_normal_ids(float* out, int64_t per_sample, int32_t nsamples, const int64_t* sample_ids, uint64_t seed,
                         const int32_t* stream_dev, int32_t stream_imm, void* stream);

/* Text projection head of the reference's EnhancedTextEncoder (text_encoder.py:13-18,31-43), applied to the
 * last_hidden_state of any text encoder (the DeBERTa weights themselves are third-party and stay outside this library):
 *   projected[b] = GELU(Linear(LayerNorm(cat(prompt_tokens, hidden[b]))))   (B, P + N0, Dt) -> xf_out
 *   pooled[b]    = mean over the P + N0 tokens of projected[b]              (B, Dt)         -> xf_proj
 * hidden fp32 (B, N0, Hs), prompts fp32 (P, Hs), Hs <= 1024.  ws >= mdm_text_head_workspace_bytes(...). */
int64_t mdm_text_head_workspace_bytes(int32_t B, int32_t N0, int32_t P, int32_t Hs, int32_t Dt);
int mdm_text_head_forward(const float* hidden, const float* prompts, const float* ln_w, const float* ln_b,
                          const MdmPacked* w, const float* bias, int32_t B, int32_t N0, int32_t P, int32_t Hs,
                          int32_t Dt, float* xf_out, float* xf_proj, void* ws, int64_t ws_bytes, int32_t precision,
                          void* stream);

/* Post-processing of generated motions on the device (tools/visualization.py:21-27,89): de-normalise (x * std + mean),
 * recover_from_ric (utils/motion_process.py:362-416: root rotation / translation by prefix sums, joints rotated back by
 * the inverse root rotation) and the temporal gaussian filter of motion_temporal_filter (utils/utils.py:125-130).
 * motion (B, T, feats) fp32, length (B) int32 or NULL (= T), mean / std (feats).  weights[0..radius]: the normalised
 * gaussian taps w[k] = w[-k] as fp64 (radius 0 = no filter).  scratch and joints_out: (B, T, joints, 3) fp32; frames
 * past a sample's length are written as zeros. */
int mdm_motion_postprocess(const float* motion, const int32_t* length, const float* mean, const float* std, int32_t B,
                           int32_t T, int32_t feats, int32_t joints, int32_t radius, const double* weights,
                           float* scratch, float* joints_out, void* stream);

/* ---- training step of the MoE feed-forward block --------------------------------------------
 * MoEMultiBranchFFN.forward (multi_branch.py:52-61) with both SwitchMoELayers (switch_moe.py:44-111) and the StylizationBlock
 * (stylization.py:20-31) in training mode, and its backward: what loss.backward() does for this block inside
 * DDPMTrainer.update (ddpm_trainer.py:228-244).  fp32 master parameters in the state_dict layouts, the two branches stacked on
 * a leading dimension; the same struct holds the gradients (same shapes, OVERWRITTEN by the backward).  Dropout
 * (multi_branch.py:57 on each branch's output, stylization.py:16 after the SiLU): dropout_p in [0, 1), masks from the
 * counter-based generator keyed on (seed, site, row, element), regenerated by the backward (pass the same p and seed);
 * p > 0 needs D in {256, 512, 1024}.  The GEMMs are the bf16x3 (fp32-grade) kernel; gradients match fp32 autograd to ~1e-5. */
typedef struct MdmMoeTensors {
  float* ln_w;      /* [2][D]        branches.{b}.layernorm.weight */
  float* ln_b;      /* [2][D] */
  float* gate_w;    /* [2][E][D]     branches.{b}.moe.gate.weight */
  float* gate_b;    /* [2][E] */
  float* w1;        /* [2][E][F][D]  branches.{b}.moe.experts.{e}.0.weight */
  float* b1;        /* [2][E][F] */
  float* w2;        /* [2][E][D][F]  branches.{b}.moe.experts.{e}.2.weight */
  float* b2;        /* [2][E][D] */
  float* st_emb_w;  /* [2D][Te]      proj_out.emb_layers.1.weight */
  float* st_emb_b;  /* [2D] */
  float* st_norm_w; /* [D]           proj_out.norm */
  float* st_norm_b;
  float* st_out_w;  /* [D][D]        proj_out.out_layers.2.weight */
  float* st_out_b;  /* [D] */
} MdmMoeTensors;

int64_t mdm_moe_train_workspace_bytes(int32_t B, int32_t S, int32_t D, int32_t F, int32_t E, int32_t Te);
/* out = x + proj_out(mean_b moe_b(LN_b(x)), emb): x [B*S, D], emb [B, De]; when De != Te the captured per-call projection
 * eph_w [Te, De], eph_b [Te] of stylization.py:22-24 is applied first (not trained).  lb_loss (optional, device [2]):
 * get_load_balancing_loss (switch_moe.py:113-145) of the two layers from this forward's counters.  route_out (optional):
 * the top-2 decisions, int32 [2][B*S][2].  Activations needed by the backward stay in ws. */
int mdm_moe_ffn_train_forward(const MdmMoeTensors* params, int32_t D, int32_t F, int32_t E, int32_t Te, int32_t De,
                              const float* eph_w, const float* eph_b, const float* x, const float* emb, int32_t B, int32_t S,
                              float dropout_p, uint64_t seed, float* out, float* lb_loss, int32_t* route_out, void* ws,
                              int64_t ws_bytes, void* stream);
/* given dout = dL/dout [B*S, D] and the workspace of the matching forward: dx [B*S, D], demb [B, De] (optional) and every
 * parameter gradient in `grads`. */
int mdm_moe_ffn_train_backward(const MdmMoeTensors* params, int32_t D, int32_t F, int32_t E, int32_t Te, int32_t De,
                               const float* eph_w, const float* x, const float* emb, int32_t B, int32_t S, float dropout_p,
                               uint64_t seed, const float* dout, float* dx, float* demb, const MdmMoeTensors* grads, void* ws,
                               int64_t ws_bytes, void* stream);
/* optimizer plumbing of ddpm_trainer.py:228-244 on flat fp32 buffers: squared gradient norm (device scalar, for
 * clip_grad_norm_) and one Adam step with the clip factor min(1, max_norm / (sqrt(*sumsq) + 1e-6)) folded in (sumsq NULL or
 * max_norm <= 0: no clip). */
int mdm_sumsq(const float* x, int64_t n, float* out, void* stream);
int mdm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  int32_t step, const float* sumsq, float max_norm, void* stream);

/* small helpers used by the host module */
int mdm_xattn_gate(const float* gate, const float* adaptive_gate, int32_t D, float* out, void* stream);
int mdm_fill_i64(int64_t* dst, int64_t n, const int32_t* src_dev, void* stream);
int mdm_add_i32(int32_t* dst, int32_t delta, void* stream);

/* Kernel-selection knob for same-box A/B runs and tests (0 = default; process-global, not thread-safe, never part of the data
 * path).  Values: 1 / 2 force the 128- / 64-row tile of the 16-bit GEMM, 6 / 7 force / forbid its 256 x 256 tile, 28 two-stage
 * ring in the 64-row tile; 21 expert MLP as two GEMMs instead of the fused kernel, 34 the LDS-staged fused kernel (csrc/mlp.hip)
 * instead of the streamed-weight one (csrc/mlp_stream.hip); 32 / 33 the Performer proj_out pair / the 4x FFN pair as two GEMMs; 30 stylization input and
 * its Linear as two launches, 29 that kernel on 64-row tiles, 35 the Performer tail as its own launches instead of inside the proj_out
 * pair's launch, 50 the Performer's q | k | v projection as its own GEMM launch instead of inside the attention core's, 51 the same for
 * the query of the linear cross-attention; 22 unfolded text cross-attention, 24 folded at any pass count;
 * 23 generic head_dim-256 paths; 25 fp32 instead of 16-bit intermediates; 26 / 27 router with compile-time / run-time expert
 * count wherever both exist; 31 input embedding in the mode's own precision; 36 fp32-grade Linears on the register-staged
 * kernel, 37-39 ring depths of the LDS-DMA fp32-grade kernel; fp32-grade fusions: 52 attention chain / 56 cross-attention chains as
 * separate launches, 58 the fused stylization on 64-row tiles, 60 stylization input and its Linear as two launches, 61 the
 * LayerNorms / block tails behind it as their own launches, 62 fp32 rows instead of pre-split rows between the bf16x3 GEMMs;
 * streamed-weight GEMM (MdmGemmDesc.w_stream): 63 never, 68 wherever eligible, 64-67 the same with a forced tile shape (112 x 512,
 * 64 x 512, 64 x 256, 32 x 256); its bf16x3 form (pre-split rows x a pair stream): 69 never, 70 wherever eligible.
 * 41-49 and 74-77 (timing-only knock-outs and the stamped build of the fused expert MLP, knock-outs of the fused stylization
 * launch: outputs are WRONG under them) exist only in the
 * diagnostic library (-DMDM_DIAG: `python motiondiffusion-moe_amd/build.py --diag` -> libmdm_hip_diag.so, used by tools/mlp_ko.py
 * and tools/mlp_stamps.py); libmdm_hip.so returns MDM_ERR_ARG for them and leaves the knob unchanged. */
int mdm_set_gemm_variant(int variant);
/* 1 in the diagnostic library, 0 in the product library */
int mdm_diag_build(void);
/* diagnostic library only (MDM_ERR_UNSUPPORTED otherwise): the eight 64-bit device counters that the stamped build of the fused
 * expert MLP (knob 49) adds its per-phase cycle sums to; NULL detaches them (knob 49 is then refused with MDM_ERR_ARG). */
int mdm_diag_mlp_counters(uint64_t* dev_counters8);
/* diagnostic: s_memtime stamps of block 0 of the last bf16 GEMM launched with feat_S == -77 (host copy, synchronises) */
int mdm_debug_stamps(uint64_t* out16);

/* Measurement probe for bench.py: while enabled, every launch of the dominant kernel (the fused expert MLP inside
 * mdm_denoiser_forward / mdm_block_forward) is bracketed by a pair of HIP events recorded on the launch stream (do not
 * enable during hipGraph capture).  mdm_probe_read synchronises the events and returns the number of launches recorded
 * since the last enable, writing up to `cap` durations (microseconds) and row counts. */
int mdm_probe_enable(int32_t enable);
/* Test aid: while buf is non-NULL, mdm_denoiser_forward copies the router's decisions into it, laid out like
 * forced_routing: int32 [2L][2 branches][B*S_layer (padded to B*T)][2] (mdm_block_forward(MDM_BLOCK_MOE): one layer's worth).
 * capacity = int32 elements buf holds: a forward that needs more (2L * 4 * B * T) returns MDM_ERR_ARG instead of writing past it.
 * Used to count routing flips against the oracle; pass NULL to switch it off.  Process-global, not thread-safe. */
int mdm_route_dump(int32_t* buf, int64_t capacity);
/* Number of passes (of <= 128 folded text columns, whole heads) the fused text cross-attention takes for H heads and N text
 * tokens, 0 when the folded path is not taken (D != 512, or more than two passes: N > 64 at H = 4, where the GEMM chain
 * measures faster).  Sizes the optional MdmTextCache buffers:
 * sd_kfold [2L][B][passes][128][D] (16-bit), sd_cb [2L][B][passes][128] (fp32), sd_vfold [2L][B][passes][D][128] (16-bit). */
int mdm_sd_fold_passes(int32_t D, int32_t H, int32_t N);
int mdm_probe_read(float* us, int32_t* rows, int32_t cap);

const char* mdm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MDM_HIP_H */
