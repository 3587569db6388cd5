// Fused MFMA GEMM family:  C = epilogue( A[M,K] * W[N,K]^T )  on gfx950.
// One kernel template covers dense Linear layers, the Performer/cross-attention contractions
// (batched, operands read straight out of activation tensors) and the grouped expert GEMMs.
// The descriptor is the public C struct (include/mdm_hip.h).
#pragma once
#include "mdm_common.h"
#include "mdm_hip.h"

namespace mdm {

typedef MdmOperand Operand;
typedef MdmGemmDesc GemmArgs;

enum OperandKind { OP_F32_ROW = MDM_OP_F32_ROW, OP_F32_KSTRIDE = MDM_OP_F32_KSTRIDE, OP_BF16_ROW = MDM_OP_BF16_ROW, OP_X2_ROW = MDM_OP_X2_ROW };
enum Act { ACT_NONE = MDM_ACT_NONE, ACT_GELU = MDM_ACT_GELU, ACT_SILU = MDM_ACT_SILU, ACT_FEAT = MDM_ACT_FEAT, ACT_HEADNORM = MDM_ACT_HEADNORM, ACT_HEADSOFTMAX = MDM_ACT_HEADSOFTMAX };

inline GemmArgs gemm_defaults(int precision) {
  GemmArgs g = {};
  g.batch = 1;
  g.nb2 = 1;
  g.alpha = 1.f;
  g.out_scale = 1.f;
  g.r1_scale = 1.f;
  g.precision = precision;
  g.a_scale_u = 1.f;
  g.c8_scale = 1.f;
  return g;
}
inline Operand op_f32(const float* p, int64_t ld) {
  Operand o = {};
  o.p = p;
  o.ld = ld;
  o.kind = OP_F32_ROW;
  return o;
}
inline Operand op_f32_kstride(const float* p, int64_t ld) {
  Operand o = {};
  o.p = p;
  o.ld = ld;
  o.kind = OP_F32_KSTRIDE;
  return o;
}
inline Operand op_bf16(const uint16_t* hi, const uint16_t* lo, int64_t ld) {
  Operand o = {};
  o.p = hi;
  o.p_lo = lo;
  o.ld = ld;
  o.kind = OP_BF16_ROW;
  return o;
}

int gemm(const GemmArgs& a, hipStream_t stream);
bool gemm_bf16_eligible(const GemmArgs& a);
int gemm_bf16(const GemmArgs& a, hipStream_t stream);  // gemm2.hip: bf16 x bf16 throughput kernel
bool gemm_bf16_256_eligible(const GemmArgs& a);
int gemm_bf16_256(const GemmArgs& a, hipStream_t stream);  // gemm4.hip: 256x256 tile, 8 waves
bool fused_mlp_supported(const MdmMlpDesc& a);
int fused_mlp(const MdmMlpDesc& a, hipStream_t stream);  // mlp.hip: Linear-GELU-Linear, hidden layer kept in LDS
bool gemm_x3_dma_eligible(const GemmArgs& a);
int gemm_x3_dma(const GemmArgs& a, hipStream_t stream);  // gemm3.hip: bf16x3 with LDS-DMA staged fp32 activations
bool gemm_fp8_eligible(const GemmArgs& a);
int gemm_fp8(const GemmArgs& a, hipStream_t stream);  // gemm8.hip: e4m3 operands, block-scaled MFMA K = 128
// gemm_stream.hip: plain Linear on 16-bit rows with the weight as a fragment stream (GemmArgs.w_stream)
int64_t gemm_stream1_elems(int N, int K);
int gemm_stream1_pack(const float* w, int64_t ldw, int N, int K, int h16, uint16_t* out, hipStream_t stream);
bool gemm_stream1_eligible(const GemmArgs& a);
bool gemm_stream1_wanted(const GemmArgs& a);  // eligible and measured faster than the tile kernel at this shape / epilogue
int gemm_stream1(const GemmArgs& a, hipStream_t stream);
// gemm_stream3.hip: the same in bf16x3 on pre-split rows, dense or grouped + gathered (the fp32-grade expert GEMMs)
int64_t gemm_stream3x_elems(int G, int N, int K);
int64_t gemm_stream3x_group_elems(int N, int K);
int gemm_stream3x_pack(const float* w, int64_t ldw, int G, int N, int K, uint16_t* out, hipStream_t stream);
bool gemm_stream3x_eligible(const GemmArgs& a);
bool gemm_stream3x_wanted(const GemmArgs& a);  // eligible and measured faster than the tile kernel of gemm3.hip
int gemm_stream3x(const GemmArgs& a, hipStream_t stream);
// mlp_stream.hip: the same MLP with the weights streamed global -> registers from a packed fragment stream
bool fused_mlp_stream_supported(const MdmMlpDesc& a);
int fused_mlp_stream(const MdmMlpDesc& a, hipStream_t stream);
// the Performer tail behind the proj_out pair, in the pair's launch (csrc/mlp_stream.hip: pair_tail)
struct PairTail {
  const float *pw, *pb;   // post_norm
  const float *sw, *sb;   // style norm
  const float* sc;        // (B, 2 D) scale | shift
  int S;                  // frames per sample
  const uint16_t* ws;     // weight stream of out_layers.2 (mdm_gemm_stream_pack)
  const float* bias;
  const float* resid;     // [M, D]
  float out_scale;
  float* out;             // fp32 [M, D]
  const float *lw, *lb;   // optional LayerNorm of the output rows ...
  uint16_t* ln16;         // ... written here as 16-bit rows
  // optional block tail of DualSelfAttentionBlock (fast_attention.py:219-225) when this is its second Performer: with
  // r = resid + out_scale * style(...) the row becomes out = LN(skip + skip_scale * r; lw, lb) as fp32 (r itself is not written),
  // and ln16 = LN(out; l2w, l2b) when l2w is set (the pre-norm of the block that follows)
  const float* skip;      // [M, D] fp32, or NULL
  float skip_scale;
  const float *l2w, *l2b;
};
bool fused_pair_style_supported(const MdmMlpDesc& a);
int fused_pair_style(const MdmMlpDesc& a, const PairTail& t, hipStream_t stream);
int64_t mlp_stream_elems(int G, int F, int Din, int Dout);
int mlp_stream_pack(const float* w1, const float* w2, int G, int F, int Din, int Dout, int h16, uint16_t* out, hipStream_t stream);
// style_gemm.hip: stylization input + its D x D Linear + the residual in one launch (streamed weights)
int64_t gemm_stream_elems(int N, int K);
int gemm_stream_pack(const float* w, int N, int K, int h16, uint16_t* out, hipStream_t stream);
bool style_gemm_supported(int D, int64_t M);
// fp32-grade form: fp32 source rows, bf16x3 products, (hi, lo) fragment pair stream
int64_t gemm_stream3_elems(int N, int K);
int gemm_stream3_pack(const float* w, int N, int K, uint16_t* out, hipStream_t stream);
// what style_gemm3 may do with the finished row r while it is on chip (all optional; see StyleGemmArgs in csrc/style_gemm.hip)
struct StyleTail3 {
  const float *lw = nullptr, *lb = nullptr;  // skip == NULL: ln_out = LN(r; lw, lb);  else out = LN(skip + skip_scale * r; lw, lb)
  float* ln_out = nullptr;                   // fp32 [M, D]
  const float* skip = nullptr;               // [M, D] fp32
  float skip_scale = 0.f;
  const float *l2w = nullptr, *l2b = nullptr;  // with skip: ln_out = LN(out; l2w, l2b)
  int ln_x2 = 0;                               // ln_out as MDM_OP_X2_ROW rows instead of fp32
};
int style_gemm3(const float* src, int64_t M, int D, int S, const float* pw, const float* pb, const float* sw, const float* sb,
                const float* sc, const int* pos4, const uint16_t* ws3, const float* bias, const float* resid, float out_scale,
                const float* colscale, float* out, const StyleTail3& t, hipStream_t s);
int style_gemm(const void* src, int src_fmt, int64_t M, int D, int S, const float* pw, const float* pb, const float* sw, const float* sb,
               const float* sc, const int* pos4, const uint16_t* ws, const float* bias, const float* resid, float out_scale,
               const float* colscale, float* out, uint16_t* out16, int h16, hipStream_t s);

}  // namespace mdm
