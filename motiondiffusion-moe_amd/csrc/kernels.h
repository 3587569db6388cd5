// Internal launchers of the non-GEMM kernels (rowwise.hip).  All are stream-ordered, allocation free.
#pragma once
#include "mdm_common.h"

namespace mdm {

// rows of the fp32 schedule table handed to the sampler kernels: tab[row * steps + t]
enum { TAB_SQRT_RECIP = 0, TAB_SQRT_RECIPM1 = 1, TAB_COEF1 = 2, TAB_COEF2 = 3, TAB_LOGVAR = 4, TAB_ACP = 5,
       TAB_ACP_PREV = 6, TAB_ROWS = 7 };

struct MoeGateParams {
  const float* ln_w[2];
  const float* ln_b[2];
  const float* gate_w[2];  // (E, D) fp32
  const float* gate_b[2];  // (E)
  void* hn;                // (2, M, D) out: LN_b(x), fp32 or bf16
  int hn_bf16;             // 0 = fp32, 1 / 2 = the 16-bit format code (MDM_H16_*), 3 = fp8 e4m3 rows + hn_scale
  float* hn_scale;         // (2, M) per-row scales of the fp8 rows (amax / 448)
  int* top_idx;            // (2, M, 2)
  float* top_val;          // (2, M, 2)
  int* hist;               // [1024][32] per-block partial histograms (no atomics, no memset)
  float* uimp;             // [1024][64] per-block partial usage | importance increments
  float* usage[2];         // optional persistent counters (E) each (switch_moe.py:71-92)
  float* importance[2];
  const int* forced_idx;   // optional (2, M, 2) injected routing (tests)
};

// y1/y2/out: fp32 tensors, or 16-bit (uint16_t) when the matching *_bf flag is non-zero; the flag IS the format code
// (MDM_H16_BF16 = 1, MDM_H16_F16 = 2).  x_bf: the input rows are 16-bit in that format (tensors whose only consumer is
// this LayerNorm)
int ln_chain(const float* x, int64_t M, int D, const float* w1, const float* b1, void* y1, int y1_bf, const float* w2,
             const float* b2, void* y2, int y2_bf, hipStream_t s, int x_bf = 0);
int style_in(const float* x, int64_t M, int D, int S, const float* pw, const float* pb, const float* sw,
             const float* sb, const float* sc, const int* pos4, int x_bf, void* out, int out_bf, hipStream_t s);
int to_bf16(const float* src, int64_t n, uint16_t* dst, int h16, hipStream_t s);  // h16: MDM_H16_BF16 / MDM_H16_F16
int moe_route(const float* x, int64_t M, int D, int E, const MoeGateParams& p, int* goff, int* cursor, int* perm,
              float* rowscale, int* pos4, hipStream_t s);
int head_norm(float* qkv, int64_t M, int H, int dh, const float* w, const float* b, hipStream_t s);
int den_ln(const float* num, const float* phi, int64_t M, int H, int dh, const float* w, const float* b, void* out,
           int out_bf, hipStream_t s);
int head_softmax(float* q, int64_t units, int dh, hipStream_t s);
// fused Performer attention core (perf_attn.hip): qkv fp32 (M,3D) -> LN_dh(num/den) as bf16 (M,D)
bool perf_attn_supported(int dh, int S);
bool perf_attn_qkv_supported(int dh, int S, int H);
int perf_attn_qkv(const uint16_t* xn, const uint16_t* wqkv, int ldw, const float* bias, float alpha, uint16_t* qscratch, int h16,
                  const uint16_t* PT, int ldp, const float* hn_w, const float* hn_b, const int* len, int B, int S, int H, int dh,
                  uint16_t* out, hipStream_t s);
int perf_attn(const void* qkv, int h16, const uint16_t* PT, int ldp, const float* hn_w, const float* hn_b, const int* len, int B,
              int S, int H, int dh, uint16_t* out, hipStream_t s);
// perf_attn2.hip: the same core at head_dim 256 (big model) in two launches (feature maps; KV state + num + LN); scratch:
// qphi | kphi^T | den, perf_attn256_scratch_bytes()
bool perf_attn256_supported(int dh, int S);
int64_t perf_attn256_scratch_bytes(int B, int H, int S);
int perf_attn256(const void* qkv, int h16, const uint16_t* PT, int ldp, const float* hn_w, const float* hn_b, const int* len,
                 int B, int S, int H, uint16_t* out, void* scratch, hipStream_t s);
// perf_attn3.hip: the same core in the fp32-grade (bf16x3) arithmetic, head_dim 128: xh / xl = hi / lo planes [B S, 3 D] of the
// q | k | v rows after LayerNorm(dh) and the L2 normalisation of q, k (the projection GEMM's ACT_HEADNORM epilogue), ph / pl =
// planes of P^T [128][ldp]; out fp32 [B S, D]
bool perf_attn3_supported(int dh, int S);
// out_x2: the output rows pre-split for the GEMM that follows (MDM_OP_X2_ROW) instead of fp32
int perf_attn3(const uint16_t* xh, const uint16_t* xl, const uint16_t* ph, const uint16_t* pl, int ldp, const float* hn_w,
               const float* hn_b, const int* len, int B, int S, int H, int dh, float* out, int out_x2, hipStream_t s);
// xattn3.hip: the text cross-attention cores of the fp32-grade modes (bf16x3 products), head_dim 128.  qh / ql: hi / lo planes
// [B S, D] of the query projection (after the head_dim softmax for lin_xattn3: ACT_HEADSOFTMAX; scaled by dh^-1/2 for sd_attn3)
bool xattn3_supported(int dh, int N);
int lin_xattn3(const uint16_t* qh, const uint16_t* ql, const float* at, int B, int S, int H, int dh, float* out, hipStream_t s);
int sd_attn3(const uint16_t* qh, const uint16_t* ql, const float* kc, const float* vc, const int32_t* ntok, int B, int S, int H,
             int dh, int N, float* out, int out_x2, hipStream_t s);
// ntok (optional, int32 [rows / rows_per_b]): row r attends to its first ntok[r / rows_per_b] columns only, the rest get 0
int row_softmax(float* sc, int64_t rows, int N, hipStream_t s, const int32_t* ntok = nullptr, int64_t rows_per_b = 1);
// fused text cross-attention cores (xattn.hip), head_dim 128
bool xattn_supported(int dh, int N);
bool lin_xattn_supported(int dh);  // linear cross-attention core: head_dim 128 or 256
int sd_attn(const void* q, int q_fmt, const float* kc, const float* vc, int B, int S, int H, int dh, int N, uint16_t* out16,
            float* out32, int h16, hipStream_t s, const int32_t* ntok = nullptr);  // ntok: per-sample token counts [B] or null
int lin_xattn(const void* ql, int ql_fmt, const float* at, int B, int S, int H, int dh, float* out, uint16_t* out16,
              int h16, hipStream_t s);
bool lin_xattn_q_supported(int dh, int S, int H);
int lin_xattn_q(const uint16_t* xn, const uint16_t* wq, int ldw, const float* bq, const float* at, int B, int S, int H, int dh, float* out,
                uint16_t* out16, int h16, hipStream_t s);
// sdfold.hip: text cross-attention with folded projections + the following LayerNorm, one launch
bool sd_fold_supported(int D, int H, int N);
int sd_fold_heads_per_pass(int H, int N);  // whole heads per pass of <= 128 folded columns
int sd_fold_passes(int H, int N);
int sd_fold(const uint16_t* x16, const uint16_t* kfold, const float* cb, const uint16_t* vfold, const float* bout,
            const float* ln_w, const float* ln_b, int B, int S, int D, int H, int N, float* out32, uint16_t* out16,
            int h16, hipStream_t s);
int col_softmax(float* k, int B, int N, int D, hipStream_t s, const int32_t* ntok = nullptr);  // ntok: tokens >= ntok[b] get 0
// cb [B][np][128] += -1e30 on the folded columns hs * N + n with n >= ntok[b] (csrc/sdfold.hip: per-row text token counts)
int sd_fold_mask_cb(float* cb, int B, int np, int hpp, int N, const int32_t* ntok, hipStream_t s);
int sinusoid(const int64_t* t, int B, int D, float* out, hipStream_t s);
int gated_mix(const float* t, const float* x, int64_t n, float* out, hipStream_t s);
int gated_mix_gather(const float* table, const int64_t* ts, int steps, const float* x, int B, int D, float* out,
                     uint16_t* out16, int h16, hipStream_t s);
int text_assemble(const float* pp, const float* ph, int B, int N0, int P, int Dt, float* xf_out, float* xf_proj,
                  hipStream_t s);
// motion_post.hip: 263-d HumanML3D rows -> (T, J, 3) joints, optional temporal gaussian filter (wts[0..radius])
int motion_post(const float* x, const int* len, const float* mean, const float* sd, int B, int T, int feats, int J,
                int radius, const double* wts, float* raw, float* out, hipStream_t s);
// noise.hip: Philox4x32-10 + Box-Muller, keyed on (seed, global sample, stream, element)
int philox_normal(float* out, int64_t per_sample, int nsamples, int64_t sample0, const int64_t* sample_ids, uint64_t seed,
                  const int* stream_dev, int stream_imm, hipStream_t s);
int iota_i64(int64_t* dst, int64_t n, int64_t start, hipStream_t s);
int xattn_gate(const float* gate, const float* ag, int D, float* out, hipStream_t s);
int halve_lengths(const int* len, int B, int* out, hipStream_t s);
int fill_i64(int64_t* dst, int64_t n, const int* src, hipStream_t s);
int add_i32(int* dst, int delta, hipStream_t s);
int cfg_step(const float* x, const float* eps_c, const float* eps_u, const float* noise, int64_t n, const float* tab,
             int ts, const int* t_ptr, int t_imm, float cfg_scale, int clip, float* x_out, float* x0_out, hipStream_t s);
int ddim_step(const float* x, const float* eps, const float* noise, int64_t n, const float* tab, int ts,
              const int* t_ptr, int t_imm, float eta, int clip, float* x_out, float* x0_out, hipStream_t s);

}  // namespace mdm
