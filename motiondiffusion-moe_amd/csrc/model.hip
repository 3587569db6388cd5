// Host-side orchestration of one denoiser forward: a straight-line sequence of stream-ordered kernel
// launches (no allocation, no host sync) so that a whole sampling step can be captured into a hipGraph.
// Reference arithmetic: text2motion/models/transformer.py:291-361 and the blocks it calls (cited inline).
#include "gemm.h"
#include "kernels.h"

namespace mdm {
extern int g_bf16_variant;
namespace {

#define MDM_TRY(expr)            \
  do {                           \
    int st__ = (expr);           \
    if (st__ != MDM_OK) return st__; \
  } while (0)

// ---- measurement probe (mdm_probe_enable / mdm_probe_read): HIP events around the dominant kernel's launches -------
constexpr int PROBE_MAX = 64;
struct Probe {
  bool on = false;
  int n = 0;
  hipEvent_t a[PROBE_MAX], b[PROBE_MAX];
  int rows[PROBE_MAX];
  bool made = false;
};
Probe g_probe;
int32_t* g_route_dump = nullptr;  // mdm_route_dump: where the router's top-2 indices of every layer are copied (tests)
int64_t g_route_cap = 0;           // its capacity in int32 elements: a forward that would write past it is refused

struct Bump {  // carve the caller's workspace; with base == nullptr it only measures
  uint8_t* base;
  int64_t off = 0;
  explicit Bump(void* b) : base((uint8_t*)b) {}
  template <typename T>
  T* take(int64_t n) {
    off = (off + 255) & ~(int64_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * (int64_t)sizeof(T);
    return p;
  }
};

struct Work {
  // token-major activations (M = B*T rows).  x?16 / h016: bf16 shadows of the fp32 residual stream, written by the
  // producing GEMM's epilogue in the throughput mode so that the next GEMM streams bf16 straight into LDS.
  float *xa, *xb, *h0, *t1, *t2, *t3, *t4, *t5, *qkv, *phi, *kvt, *scr, *f1, *hn, *hid, *y2;
  uint16_t *xa16, *xb16, *h016;
  int *top_idx, *perm, *pos4, *hist, *goff, *cursor, *len_low;
  float *top_val, *rowscale, *uimp, *hn_scale;
  // stem (B rows)
  float *s_a, *s_b, *s_c, *emb, *e1, *sc, *gvtmp;
  // text cache scratch
  float *tn, *kb, *vb;
  int64_t bytes;
};

Work carve(const MdmModel& m, int B, int T, int N, void* ws) {
  Work w;
  Bump b(ws);
  const int64_t M = (int64_t)B * T, D = m.D, F = m.F, Te = 4 * D, nblk = 8 * m.L;
  const int dh = m.D / m.H;
  w.xa = b.take<float>(M * D), w.xb = b.take<float>(M * D), w.h0 = b.take<float>(M * D);
  w.xa16 = b.take<uint16_t>(M * D), w.xb16 = b.take<uint16_t>(M * D), w.h016 = b.take<uint16_t>(M * D);
  w.t1 = b.take<float>(M * D), w.t2 = b.take<float>(M * D), w.t3 = b.take<float>(M * D);
  w.t4 = b.take<float>(M * D), w.t5 = b.take<float>(M * D);
  w.qkv = b.take<float>(M * 3 * D);
  w.phi = b.take<float>(M * 2 * D);
  w.kvt = b.take<float>((int64_t)B * m.H * dh * dh);
  w.scr = b.take<float>(M * m.H * (N > 0 ? N : 1));
  w.f1 = b.take<float>(M * 4 * D);
  w.hn = b.take<float>(2 * M * D);
  w.hid = b.take<float>(4 * M * F);
  w.y2 = b.take<float>(4 * M * D);
  w.top_idx = b.take<int>(4 * M), w.top_val = b.take<float>(4 * M);
  w.perm = b.take<int>(4 * M), w.rowscale = b.take<float>(4 * M), w.pos4 = b.take<int>(4 * M);
  w.hn_scale = b.take<float>(2 * M);
  w.hist = b.take<int>(1024 * 32), w.uimp = b.take<float>(1024 * 64), w.goff = b.take<int>(2 * m.E + 1), w.cursor = b.take<int>(2 * m.E);
  w.len_low = b.take<int>(B);
  const int64_t smax = Te > 2 * D ? Te : 2 * D;
  w.s_a = b.take<float>(B * smax), w.s_b = b.take<float>(B * smax), w.s_c = b.take<float>(B * smax);
  w.emb = b.take<float>(B * D);
  w.e1 = b.take<float>(B * nblk * Te);
  w.sc = b.take<float>(nblk * B * 2 * D);
  w.gvtmp = b.take<float>(D);
  const int64_t BN = (int64_t)B * (N > 0 ? N : 1);
  w.tn = b.take<float>(BN * m.Dt), w.kb = b.take<float>(BN * D), w.vb = b.take<float>(BN * D);
  w.bytes = (b.off + 255) & ~(int64_t)255;
  return w;
}

struct Ctx {
  const MdmModel* m;
  hipStream_t s;
  int prec;         // GEMMs with fp32 activations: 1 = single bf16 pass, 3 = bf16x3
  bool bf;          // throughput modes (precision 1 / 2): GEMM-only tensors are kept in 16 bits
  int h16;          // the 16-bit format of this run: MDM_H16_BF16 (precision 1, 3) or MDM_H16_F16 (precision 2, 4)
  bool mix;         // precision 4: fp32-grade flow, but expert MLPs + the 4x FFN run as ONE fp16 pass on 16-bit operands
  bool fp8;         // precision 5: as 2, expert GEMMs on e4m3 operands (csrc/gemm8.hip)
  bool x2;          // fp32-grade GEMM chains: a tensor whose only consumer is a bf16x3 GEMM is written PRE-SPLIT by its producer
                    // (MDM_OP_X2_ROW rows: same bytes as fp32) -- the GEMM then does not re-split every fragment in its K loop (knob 62: off)
  int B, S, N;      // batch, frames at this scale, text tokens
  const int32_t* ntok = nullptr;  // per-sample text token counts [B] (MdmTextCache.ntok), or null: all N
  int64_t M;        // B*S
  const int* len;   // lengths at this scale
  Work w;
};

inline Operand packed(const MdmPacked& p) { return op_bf16(p.hi, p.lo, p.ld); }

// bf16 activation plumbing needs every GEMM K (D, 2D, 4D, F) to be a multiple of the 64-wide LDS-DMA k-tile
bool use_bf16_acts(const MdmModel* m, int precision) {
  return (precision == MDM_PREC_BF16 || precision == MDM_PREC_F16) && m->D % 64 == 0 && m->F % 64 == 0;
}
// precision (include/mdm_hip.h: MDM_PREC_*) -> how this run computes; false = unsupported combination
bool set_precision(Ctx& c, const MdmModel* m, int precision) {
  c.mix = false, c.bf = false, c.fp8 = false, c.h16 = MDM_H16_BF16;
  // (knob 36 puts the fp32-grade Linears on the register-staged kernel, which reads fp32 rows only)
  c.x2 = (precision == MDM_PREC_X3 || precision == MDM_PREC_MIXED) && g_bf16_variant != 62 && g_bf16_variant != 36 &&
         (m->D == 256 || m->D == 512 || m->D == 1024) && m->F % 32 == 0;
  switch (precision) {
    case MDM_PREC_FP8:  // the fp16 throughput mode with fp8 expert GEMMs (K = D and K = F must be multiples of 128)
      c.prec = 1, c.bf = use_bf16_acts(m, MDM_PREC_F16), c.h16 = MDM_H16_F16, c.fp8 = true;
      return c.bf && m->D % 128 == 0 && m->F % 128 == 0;
    case MDM_PREC_BF16: c.prec = 1, c.bf = use_bf16_acts(m, precision); return true;
    case MDM_PREC_F16:  // fp16 weight planes are only readable by the 16-bit-activation kernels
      c.prec = 1, c.bf = use_bf16_acts(m, precision), c.h16 = MDM_H16_F16;
      return c.bf;
    case MDM_PREC_X3: c.prec = 3; return true;
    case MDM_PREC_MIXED:
      c.prec = 3, c.h16 = MDM_H16_F16, c.mix = m->D % 64 == 0 && m->F % 64 == 0;
      return c.mix;
    default: return false;
  }
}
inline GemmArgs gd(const Ctx& c) {  // GEMM descriptor defaults of this run
  GemmArgs g = gemm_defaults(c.prec);
  g.h16 = c.h16;
  return g;
}
inline int fmt16(const Ctx& c) { return c.bf ? c.h16 : 0; }             // format code of a mode-typed tensor (0 = fp32)
inline int fmt_mlp(const Ctx& c) { return (c.bf || c.mix) ? c.h16 : 0; }  // ... of the expert-MLP / FFN operands

// a tensor that is fp32 in the fp32-grade mode and bf16 in the throughput mode, living in a float-sized buffer
struct Act {
  void* p;
  bool bf;
  bool x2 = false;  // pre-split rows (MDM_OP_X2_ROW) in an fp32-sized buffer
};
inline Act act_of(const Ctx& c, float* buf) { return Act{buf, c.bf}; }
inline Act act_x2(const Ctx& c, float* buf) { return Act{buf, c.bf, !c.bf && c.x2}; }  // a tensor its producer wrote in the mode's GEMM-input form
inline Act act_f32(const float* buf) { return Act{(void*)buf, false}; }
inline Act act_bf16(const uint16_t* buf) { return Act{(void*)buf, true}; }
inline Act act_h16(const void* buf) { return Act{(void*)buf, true}; }

struct LinOpts {
  int act = ACT_NONE;
  float alpha = 1.f, out_scale = 1.f, r1_scale = 1.f;
  const float* R1 = nullptr;
  const float* R2 = nullptr;
  const float* colscale = nullptr;
  int r1_mod = 0;
  uint16_t* outx2 = nullptr;  // fp32-grade kernel: the result as pre-split rows (MdmGemmDesc.Cx2)
};

// out32 / out16 = epilogue(A @ W^T) for a plain [M,K]x[N,K] Linear; either output may be null
int linear(const Ctx& c, Act A, int64_t M, int K, const MdmPacked& W, const float* bias, int N, float* out32,
           uint16_t* out16, const LinOpts& o = LinOpts()) {
  GemmArgs g = gd(c);
  if (A.bf) {
    g.A.p = A.p, g.A.ld = K, g.A.kind = OP_BF16_ROW;
    g.precision = 1;  // 16-bit activations: one pass of the format c.h16 (the weight was packed in it)
  } else {
    g.A = op_f32((const float*)A.p, K);
    if (A.x2) g.A.kind = OP_X2_ROW;
  }
  g.W = packed(W);
  g.w_stream = A.bf ? W.ws : nullptr;  // 16-bit rows and a fragment stream of W: the streamed-weight kernel where it is eligible (csrc/gemm_stream.hip)
  g.M = (int)M, g.N = N, g.K = K;
  g.C = out32, g.C16 = out16, g.ldc = N;
  g.Cx2 = o.outx2;
  g.bias = bias;
  g.act = o.act, g.alpha = o.alpha, g.out_scale = o.out_scale;
  g.R1 = o.R1, g.ldr1 = N, g.r1_scale = o.r1_scale, g.r1_mod = o.r1_mod;
  g.R2 = o.R2, g.ldr2 = N;
  g.colscale = o.colscale;
  if (A.bf && !gemm_bf16_eligible(g)) return MDM_ERR_UNSUPPORTED;
  return gemm(g, c.s);
}
// result typed like the mode: fp32 buffer in the fp32-grade mode, bf16 in the throughput mode
int linear_to_act(const Ctx& c, Act A, int64_t M, int K, const MdmPacked& W, const float* bias, int N, float* buf,
                  const LinOpts& o = LinOpts()) {
  return linear(c, A, M, K, W, bias, N, c.bf ? nullptr : buf, c.bf ? (uint16_t*)buf : nullptr, o);
}

// out = resid + out_scale * colscale * Lin(SiLU(LN(a)(1+scale)+shift)) with a = [post-processed] src
// t3 / t3_done: fp32-grade modes only -- what the fused launch may do with the finished rows (csrc/gemm.h StyleTail3); *t3_done says
// whether it did (false: the caller runs those LayerNorms itself)
int style_apply(const Ctx& c, const MdmStyle& st, const float* src, const float* pw, const float* pb, const int* pos4,
                const float* sc, float* tmp, const float* resid, float out_scale, const float* colscale, float* out,
                uint16_t* out16 = nullptr, bool src_bf16 = false, const StyleTail3* t3 = nullptr, bool* t3_done = nullptr) {
  const int D = c.m->D;
  if (c.bf && st.out_ws && g_bf16_variant != 30 && style_gemm_supported(D, c.M))  // one launch (csrc/style_gemm.hip); knob 30: two
    return style_gemm(src, src_bf16 ? c.h16 : 0, c.M, D, c.S, pw, pb, st.norm_w, st.norm_b, sc, pos4, st.out_ws, st.out_b, resid,
                      out_scale, colscale, out, out16, c.h16, c.s);
  // fp32-grade modes: the same fusion on bf16x3 products (csrc/style_gemm.hip style_gemm3; knob 60: two launches)
  if (!c.bf && c.prec == 3 && !src_bf16 && !out16 && st.out_ws3 && g_bf16_variant != 60 && style_gemm_supported(D, c.M)) {
    const bool tail = t3 && g_bf16_variant != 61;  // knob 61: the LayerNorms behind it as their own launches
    if (t3_done) *t3_done = tail;
    return style_gemm3(src, c.M, D, c.S, pw, pb, st.norm_w, st.norm_b, sc, pos4, st.out_ws3, st.out_b, resid, out_scale, colscale, out,
                       tail ? *t3 : StyleTail3(), c.s);
  }
  MDM_TRY(style_in(src, c.M, D, c.S, pw, pb, st.norm_w, st.norm_b, sc, pos4, src_bf16 ? c.h16 : 0, tmp, fmt16(c), c.s));
  LinOpts o;
  o.out_scale = out_scale, o.R1 = resid, o.colscale = colscale;
  return linear(c, act_of(c, tmp), c.M, D, st.out, st.out_b, D, out, out16, o);
}

// PerformerSelfAttention (fast_attention.py:137-179): xn = pre_norm(x) already computed; out = x + 0.1*style(...)
// What the fused tail (csrc/mlp_stream.hip pair_tail) may do beyond the Performer itself when it runs (*done says whether it did):
//   skip == NULL: ln16 = LN(out; lw, lb)                          (the pre-norm of the block that follows)
//   skip != NULL: out  = LN(skip + skip_scale * r; lw, lb) with r the Performer's own output, which is not written;
//                 ln16 = LN(out; l2w, l2b) when l2w is set         (the tail of DualSelfAttentionBlock, fast_attention.py:219-225)
struct PerfTail {
  const float *lw = nullptr, *lb = nullptr;
  uint16_t* ln16 = nullptr;
  float* ln32 = nullptr;  // the fp32-grade modes' form of ln16 (fp32 rows; csrc/style_gemm.hip style_gemm3)
  bool ln_x2 = false;     // ... as pre-split rows for the GEMM that reads them
  const float* skip = nullptr;
  float skip_scale = 0.f;
  const float *l2w = nullptr, *l2b = nullptr;
  bool* done = nullptr;
};

// the proj_out pair (fast_attention.py:121-126) as one launch of the streamed-weight kernel, in place on t4
MdmMlpDesc proj_pair_desc(const Ctx& c, const MdmPerformer& p) {
  const int D = c.m->D;
  MdmMlpDesc f = {};
  f.X = (const uint16_t*)c.w.t4, f.ldx = D, f.M = (int)c.M, f.Din = D, f.F = D, f.Dout = D;
  f.b1 = p.proj0_b, f.b2 = p.proj3_b, f.wstream = p.proj_ws, f.wstream_gs = 2 * (int64_t)D * D;
  f.r1_scale = 1.f, f.C16 = (uint16_t*)c.w.t4, f.ldc = D, f.h16 = c.h16;  // in place: a tile's rows are in LDS before its stores
  return f;
}
// whether performer() runs that pair (knob 32: two GEMMs) / the pair AND the Performer's tail (knob 35: the tail as its own launches)
bool proj_pair_fused(const Ctx& c, const MdmPerformer& p) {
  return c.bf && g_bf16_variant != 25 && p.proj_ws && g_bf16_variant != 32;
}
bool performer_tail_fused(const Ctx& c, const MdmPerformer& p) {
  return proj_pair_fused(c, p) && p.style.out_ws && g_bf16_variant != 35 && g_bf16_variant != 30 &&
         fused_pair_style_supported(proj_pair_desc(c, p));
}

int performer(const Ctx& c, const MdmPerformer& p, const float* x, Act xn, const float* sc, float* out,
              const PerfTail& pt = PerfTail()) {
  const MdmModel& m = *c.m;
  const int D = m.D, H = m.H, dh = D / H, mf = dh;  // m = min(dh, 256) = dh for dh <= 256
  const Work& w = c.w;
  const bool fused256 = c.bf && g_bf16_variant != 23 && perf_attn256_supported(dh, c.S) &&
                        perf_attn256_scratch_bytes(c.B, H, c.S) <= c.M * 2 * D * (int64_t)sizeof(float);
  const bool fused = fused256 || (c.bf && perf_attn_supported(dh, c.S));
  // head_dim 128, 16-bit modes: the q | k | v projection inside the attention core's launch (csrc/perf_attn.hip phase 0: one
  // workgroup per (batch, head) multiplies its sample's rows with its head's 384 weight rows; k and v never leave the CU).
  // Knob 50: the projection as its own GEMM launch.
  const bool qkv_in = fused && !fused256 && xn.bf && g_bf16_variant != 50 && perf_attn_qkv_supported(dh, c.S, H);
  // fp32-grade modes, head_dim 128: the projection's epilogue applies LN(dh) / L2 and writes bf16 hi | lo planes, ONE launch
  // (csrc/perf_attn3.hip) does features -> KV state -> num / den -> LN on bf16x3 products.  Knob 52: the five-launch chain.
  const bool fused3 = !c.bf && c.prec == 3 && !xn.bf && g_bf16_variant != 52 && perf_attn3_supported(dh, c.S) && D % 32 == 0 &&
                      p.qkv.lo && p.feat.lo && (c.M * 3 * D) % 8 == 0;
  if (fused3) {
    uint16_t* const xh = (uint16_t*)w.qkv;
    uint16_t* const xl = xh + c.M * 3 * D;  // the two 16-bit planes fill the fp32 [M, 3 D] buffer exactly
    GemmArgs g = gd(c);
    g.A = op_f32((const float*)xn.p, D);
    if (xn.x2) g.A.kind = OP_X2_ROW;
    g.W = packed(p.qkv);
    g.M = (int)c.M, g.N = 3 * D, g.K = D;
    g.bias = p.qkv_b, g.alpha = 0.1f;
    g.act = ACT_HEADNORM, g.hn_w = p.hn_w, g.hn_b = p.hn_b, g.hn_l2_tiles = 2 * H;
    g.C16 = xh, g.C16_lo = xl, g.ldc = 3 * D;
    MDM_TRY(gemm(g, c.s));
    MDM_TRY(perf_attn3(xh, xl, p.feat.hi, p.feat.lo, (int)p.feat.ld, p.hn_w, p.hn_b, c.len, c.B, c.S, H, dh, w.t4, c.x2 ? 1 : 0, c.s));
  }
  const bool t4x2 = fused3 && c.x2;  // the attention rows (t4) and the projection's hidden rows (t2) travel pre-split
  // q|k|v = 0.1 * (xn W^T + b)                                   (:145-157); bf16 when the fused attention core reads it
  if (!qkv_in && !fused3) {
    LinOpts o;
    o.alpha = 0.1f;
    MDM_TRY(linear(c, xn, c.M, D, p.qkv, p.qkv_b, 3 * D, fused ? nullptr : w.qkv, fused ? (uint16_t*)w.qkv : nullptr, o));
  }
  if (fused3) {
  } else if (qkv_in) {
    MDM_TRY(perf_attn_qkv((const uint16_t*)xn.p, p.qkv.hi, (int)p.qkv.ld, p.qkv_b, 0.1f, (uint16_t*)w.qkv, c.h16, p.feat.hi,
                          (int)p.feat.ld, p.hn_w, p.hn_b, c.len, c.B, c.S, H, dh, (uint16_t*)w.t4, c.s));
  } else if (fused256) {
    // the same at head_dim 256 (big model) in two launches: feature maps (P^T resident), then KV state + num + LN per
    // (batch, head); qphi / kphi^T / den pass through w.phi (L2 / MALL resident)
    MDM_TRY(perf_attn256(w.qkv, c.h16, p.feat.hi, (int)p.feat.ld, p.hn_w, p.hn_b, c.len, c.B, c.S, H, (uint16_t*)w.t4, w.phi, c.s));
  } else if (fused) {
    // throughput mode: LN/L2 -> feature maps -> KV state -> num/den -> LN in ONE kernel per (batch, head)  (:44-90)
    MDM_TRY(perf_attn(w.qkv, c.h16, p.feat.hi, (int)p.feat.ld, p.hn_w, p.hn_b, c.len, c.B, c.S, H, dh, (uint16_t*)w.t4, c.s));
  } else {
    // shared LN over head_dim, L2 normalise q,k                     (:44-55)
    MDM_TRY(head_norm(w.qkv, c.M, H, dh, p.hn_w, p.hn_b, c.s));
    // feature maps 0.1*exp(clamp(z P)), keys masked past length     (:58-74): rows = (token, slot<2H)
    {
      GemmArgs g = gd(c);
      g.A = op_f32(w.qkv, dh);
      g.A.rpg = 2 * H, g.A.gstride = 3 * D;
      g.W = packed(p.feat);
      g.M = (int)(c.M * 2 * H), g.N = mf, g.K = dh;
      g.C = w.phi, g.ldc = mf;
      g.act = ACT_FEAT;
      g.feat_len = c.len, g.feat_S = c.S, g.feat_rpt = 2 * H, g.feat_kslot = H;
      MDM_TRY(gemm(g, c.s));
    }
    // KV^T[b,h] (dh x m) = 0.1 * sum_t v[t] (x) kphi[t]              (:77)
    {
      GemmArgs g = gd(c);
      g.A = op_f32_kstride(w.qkv + 2 * D, 3 * D);
      g.A.bs1 = (int64_t)c.S * 3 * D, g.A.bs2 = dh;
      g.W = op_f32_kstride(w.phi + H * mf, 2 * H * mf);
      g.W.bs1 = (int64_t)c.S * 2 * H * mf, g.W.bs2 = mf;
      g.M = dh, g.N = mf, g.K = c.S;
      g.batch = c.B * H, g.nb2 = H;
      g.C = w.kvt, g.ldc = mf, g.c_bs1 = (int64_t)H * dh * mf, g.c_bs2 = (int64_t)dh * mf;
      g.out_scale = 0.1f;
      MDM_TRY(gemm(g, c.s));
    }
    // num = 0.1 * qphi KV                                            (:78) -> t2 (M, D) merged heads
    {
      GemmArgs g = gd(c);
      g.A = op_f32(w.phi, 2 * H * mf);
      g.A.bs1 = (int64_t)c.S * 2 * H * mf, g.A.bs2 = mf;
      g.W = op_f32(w.kvt, mf);
      g.W.bs1 = (int64_t)H * dh * mf, g.W.bs2 = (int64_t)dh * mf;
      g.M = c.S, g.N = dh, g.K = mf;
      g.batch = c.B * H, g.nb2 = H;
      g.C = w.t2, g.ldc = D, g.c_bs1 = (int64_t)c.S * D, g.c_bs2 = dh;
      g.out_scale = 0.1f;
      MDM_TRY(gemm(g, c.s));
    }
    // same-t denominator, divide, LN over head_dim                   (:81-90) -> t4
    MDM_TRY(den_ln(w.t2, w.phi, c.M, H, dh, p.hn_w, p.hn_b, w.t4, fmt16(c), c.s));
  }
  // proj_out: Linear -> GELU -> Linear                             (:121-126,165)
  const bool t16 = c.bf && g_bf16_variant != 25;  // the projection's output feeds a LayerNorm only: bf16 in throughput mode
  bool pair = false;
  if (proj_pair_fused(c, p)) {  // one launch, hidden layer on chip (csrc/mlp_stream.hip)
    const MdmMlpDesc f = proj_pair_desc(c, p);
    // ... and the tail (post_norm, stylization, out_layers.2, residual, the next block's LayerNorm) in the same launch: the
    // pair's rows never leave the CU (knob 35: the tail as its own launch)
    if (performer_tail_fused(c, p)) {
      PairTail t = {};
      t.pw = p.post_w, t.pb = p.post_b, t.sw = p.style.norm_w, t.sb = p.style.norm_b, t.sc = sc, t.S = c.S;
      t.ws = p.style.out_ws, t.bias = p.style.out_b, t.resid = x, t.out_scale = 0.1f, t.out = out;
      t.lw = pt.lw, t.lb = pt.lb, t.ln16 = pt.ln16, t.skip = pt.skip, t.skip_scale = pt.skip_scale, t.l2w = pt.l2w, t.l2b = pt.l2b;
      MDM_TRY(fused_pair_style(f, t, c.s));
      if (pt.done) *pt.done = true;
      return MDM_OK;
    }
    if (fused_mlp_stream_supported(f)) {
      MDM_TRY(fused_mlp_stream(f, c.s));
      pair = true;
    }
  }
  if (!pair && !c.bf && c.x2) {
    LinOpts o;
    o.act = ACT_GELU, o.outx2 = (uint16_t*)w.t2;
    MDM_TRY(linear(c, Act{w.t4, false, t4x2}, c.M, D, p.proj0, p.proj0_b, D, nullptr, nullptr, o));
    MDM_TRY(linear(c, Act{w.t2, false, true}, c.M, D, p.proj3, p.proj3_b, D, w.t4, nullptr));
  } else if (!pair) {
    LinOpts o;
    o.act = ACT_GELU;
    MDM_TRY(linear_to_act(c, act_of(c, w.t4), c.M, D, p.proj0, p.proj0_b, D, w.t2, o));
    MDM_TRY(linear(c, act_of(c, w.t2), c.M, D, p.proj3, p.proj3_b, D, t16 ? nullptr : w.t4, t16 ? (uint16_t*)w.t4 : nullptr));
  }
  // post_norm, normalize * sqrt(D), stylization, y = x + 0.1 * style (:169-178); fp32-grade modes: the caller's LayerNorms / block
  // tail behind it in the same launch when it asks for them (pt.ln32 / pt.skip)
  if (!c.bf && (pt.ln32 || pt.skip)) {
    StyleTail3 t3;
    t3.lw = pt.lw, t3.lb = pt.lb, t3.ln_out = pt.ln32, t3.skip = pt.skip, t3.skip_scale = pt.skip_scale, t3.l2w = pt.l2w, t3.l2b = pt.l2b;
    t3.ln_x2 = pt.ln_x2;
    bool did = false;
    MDM_TRY(style_apply(c, p.style, w.t4, p.post_w, p.post_b, nullptr, sc, w.t2, x, 0.1f, nullptr, out, nullptr, t16, &t3, &did));
    if (pt.done) *pt.done = did;
    return MDM_OK;
  }
  return style_apply(c, p.style, w.t4, p.post_w, p.post_b, nullptr, sc, w.t2, x, 0.1f, nullptr, out, nullptr, t16);
}

// DualSelfAttentionBlock (fast_attention.py:208-226): x -> out.  Uses t1..t5.  x16: bf16 shadow of x (throughput mode)
// next_w / next_b: optional LayerNorm of the FOLLOWING block (cross_attn.base_ca.norm) chained onto post_norm; its output
// goes to t2 typed like the mode, and cross_block(pre_normed = true) picks it up (one launch less per layer)
int dual_block(const Ctx& c, const MdmLayer& l, const float* x, const uint16_t* x16, const float* sc4, float* out,
               const float* next_w = nullptr, const float* next_b = nullptr) {
  const int D = c.m->D;
  const Work& w = c.w;
  const int64_t scs = (int64_t)c.B * 2 * D;
  // With the Performers' tails fused into their projection launches (16-bit modes, D = 512) the block's own tail goes there too:
  // skip = GELU(Lin(x)) is computed FIRST (into the FFN's hidden buffer, free during this block), and the second Performer's
  // launch ends with out = post_norm(skip + 0.1 * global_out) and the next block's pre-norm -- no D x D GEMM waiting behind the
  // attention chain, no LayerNorm launch behind that.
  const bool tails = performer_tail_fused(c, l.local) && performer_tail_fused(c, l.global);
  // fp32-grade modes: the same re-ordering with the tails inside the stylization launches (style_gemm3): knobs 60 / 61 undo it
  const bool tails3 = !c.bf && c.prec == 3 && g_bf16_variant != 60 && g_bf16_variant != 61 && l.local.style.out_ws3 &&
                      l.global.style.out_ws3 && style_gemm_supported(D, c.M);
  float* const skipbuf = w.f1;
  if (tails || tails3) {
    LinOpts o;
    o.act = ACT_GELU;
    MDM_TRY(linear(c, tails ? act_bf16(x16) : act_f32(x), c.M, D, l.skip, l.skip_b, D, skipbuf, nullptr, o));
  }
  // h = pre_norm(x) -> t1 ; local.pre_norm(h) -> t3
  const int xnf = (!c.bf && c.x2) ? 4 : fmt16(c);  // format of the pre-normed rows the q | k | v projections read (4 = pre-split rows)
  MDM_TRY(ln_chain(x, c.M, D, l.dual_pre_w, l.dual_pre_b, w.t1, 0, l.local.pre_w, l.local.pre_b, w.t3, xnf, c.s));
  bool normed = false;  // local_out -> t5; the fused tail also leaves global.pre_norm(local_out) in t3 (t3 is dead by then)
  {
    PerfTail pt;
    if (c.bf) pt.lw = l.global.pre_w, pt.lb = l.global.pre_b, pt.ln16 = (uint16_t*)w.t3, pt.done = &normed;
    if (tails3) pt.lw = l.global.pre_w, pt.lb = l.global.pre_b, pt.ln32 = w.t3, pt.ln_x2 = c.x2, pt.done = &normed;
    MDM_TRY(performer(c, l.local, w.t1, act_x2(c, w.t3), sc4 + 0 * scs, w.t5, pt));
  }
  if (!normed) MDM_TRY(ln_chain(w.t5, c.M, D, l.global.pre_w, l.global.pre_b, w.t3, xnf, nullptr, nullptr, nullptr, 0, c.s));
  if ((tails || tails3) && normed) {
    bool done = false;
    PerfTail pt;
    pt.skip = skipbuf, pt.skip_scale = 0.1f, pt.lw = l.dual_post_w, pt.lb = l.dual_post_b, pt.done = &done;
    if (next_w && tails) pt.l2w = next_w, pt.l2b = next_b, pt.ln16 = (uint16_t*)w.t2;
    if (next_w && tails3) pt.l2w = next_w, pt.l2b = next_b, pt.ln32 = w.t2, pt.ln_x2 = c.x2;
    MDM_TRY(performer(c, l.global, w.t5, act_x2(c, w.t3), sc4 + 1 * scs, out, pt));
    if (done) return MDM_OK;
    return MDM_ERR_LAUNCH;  // (unreachable: performer_tail_fused said the tail runs)
  }
  MDM_TRY(performer(c, l.global, w.t5, act_x2(c, w.t3), sc4 + 1 * scs, w.t1));  // global_out -> t1
  // skip = GELU(Lin(x)); out = post_norm(skip + 0.1 * global)      (:219-225)
  {
    LinOpts o;
    o.act = ACT_GELU, o.R1 = w.t1, o.r1_scale = 0.1f;
    // fp32 on purpose: this sum is the block's output before its final LayerNorm (storing it as bf16 measured +60 % block
    // error for 0.01 ms per step)
    MDM_TRY(linear(c, c.bf ? act_bf16(x16) : act_f32(x), c.M, D, l.skip, l.skip_b, D, w.t3, nullptr, o));
  }
  return ln_chain(w.t3, c.M, D, l.dual_post_w, l.dual_post_b, out, 0, next_w, next_b, next_w ? w.t2 : nullptr, xnf, c.s);
}

// GatedCrossAttention (fast_attention.py:242-272): out = x + sigmoid(gate)*sigmoid(adaptive)*style(softmax(q) A)
int cross_block(const Ctx& c, const MdmLayer& l, const float* at, const float* x, const float* sc, float* out,
                bool pre_normed = false) {
  const MdmModel& m = *c.m;
  const int D = m.D, H = m.H, dh = D / H;
  const Work& w = c.w;
  if (!pre_normed) MDM_TRY(ln_chain(x, c.M, D, l.ca_norm_w, l.ca_norm_b, w.t2, (!c.bf && c.x2) ? 4 : fmt16(c), nullptr, nullptr, nullptr, 0, c.s));
  const bool fused = c.bf && lin_xattn_supported(dh) && !(dh == 256 && g_bf16_variant == 23);  // knob 23: big-width generic paths
  bool x16o = false;
  // head_dim 128: the query projection inside the attention launch (csrc/xattn.hip lin_xattn_q; knob 51: its own GEMM launch)
  const bool q_in = fused && g_bf16_variant != 51 && D == 512 && lin_xattn_q_supported(dh, c.S, H);
  // fp32-grade modes, head_dim 128: the head_dim softmax is the query projection's epilogue (bf16 hi | lo planes), the product with
  // A[b, h] one launch on bf16x3 MFMAs (csrc/xattn3.hip).  Knob 56: the chain (GEMM, head_softmax, batched contraction).
  const bool fused3 = !c.bf && c.prec == 3 && g_bf16_variant != 56 && xattn3_supported(dh, 1) && l.ca_q.lo && D % 32 == 0 &&
                      (c.M * D) % 8 == 0;
  if (fused3) {
    uint16_t* const qh = (uint16_t*)w.t3;
    uint16_t* const ql = qh + c.M * D;  // the two planes fill the fp32 [M, D] buffer exactly
    GemmArgs g = gd(c);
    g.A = op_f32(w.t2, D);
    if (c.x2) g.A.kind = OP_X2_ROW;
    g.W = packed(l.ca_q);
    g.M = (int)c.M, g.N = D, g.K = D;
    g.bias = l.ca_q_b, g.act = ACT_HEADSOFTMAX;
    g.C16 = qh, g.C16_lo = ql, g.ldc = D;
    MDM_TRY(gemm(g, c.s));
    MDM_TRY(lin_xattn3(qh, ql, at, c.B, c.S, H, dh, w.t4, c.s));
    return style_apply(c, l.ca_style, w.t4, nullptr, nullptr, nullptr, sc, w.t2, x, 1.f, l.ca_gvec, out, nullptr, false);
  }
  if (!q_in) MDM_TRY(linear(c, act_x2(c, w.t2), c.M, D, l.ca_q, l.ca_q_b, D, fused ? nullptr : w.t3, fused ? (uint16_t*)w.t3 : nullptr));
  if (q_in) {
    x16o = g_bf16_variant != 25;
    MDM_TRY(lin_xattn_q((const uint16_t*)w.t2, l.ca_q.hi, (int)l.ca_q.ld, l.ca_q_b, at, c.B, c.S, H, dh, x16o ? nullptr : w.t4,
                        x16o ? (uint16_t*)w.t4 : nullptr, c.h16, c.s));
  } else if (fused) {
    x16o = g_bf16_variant != 25;  // consumed by the stylization LayerNorm only: bf16
    MDM_TRY(lin_xattn(w.t3, c.h16, at, c.B, c.S, H, dh, x16o ? nullptr : w.t4, x16o ? (uint16_t*)w.t4 : nullptr, c.h16, c.s));  // (:248,253)
  } else {
    MDM_TRY(head_softmax(w.t3, c.M * H, dh, c.s));  // softmax over head_dim (:248)
    {
      GemmArgs g = gd(c);  // y[b,s,h,:] = q[b,s,h,:] A[b,h]  (:253), W = A^T rows
      g.A = op_f32(w.t3, D);
      g.A.bs1 = (int64_t)c.S * D, g.A.bs2 = dh;
      g.W = op_f32(at, dh);
      g.W.bs1 = (int64_t)H * dh * dh, g.W.bs2 = (int64_t)dh * dh;
      g.M = c.S, g.N = dh, g.K = dh;
      g.batch = c.B * H, g.nb2 = H;
      g.C = w.t4, g.ldc = D, g.c_bs1 = (int64_t)c.S * D, g.c_bs2 = dh;
      MDM_TRY(gemm(g, c.s));
    }
  }
  return style_apply(c, l.ca_style, w.t4, nullptr, nullptr, nullptr, sc, w.t2, x, 1.f, l.ca_gvec, out, nullptr, x16o);
}

// MoEMultiBranchFFN (multi_branch.py:52-61) with SwitchMoELayer top-2 routing (switch_moe.py:44-111)
int moe_block(const Ctx& c, const MdmLayer& l, const float* x, const float* sc, const int* forced, float* out,
              uint16_t* out16, int32_t* route_out = nullptr) {
  const MdmModel& m = *c.m;
  const int D = m.D, F = m.F, E = m.E;
  const Work& w = c.w;
  MoeGateParams p = {};
  for (int b = 0; b < 2; ++b) {
    p.ln_w[b] = l.moe_ln_w[b], p.ln_b[b] = l.moe_ln_b[b];
    p.gate_w[b] = l.gate_w[b], p.gate_b[b] = l.gate_b[b];
    p.usage[b] = l.usage[b], p.importance[b] = l.importance[b];
  }
  const bool hx2 = !c.bf && !c.mix && c.x2 && D % 64 == 0;  // fp32-grade experts: LN rows and hidden rows pre-split for the two GEMMs
  p.hn = w.hn, p.hn_bf16 = c.fp8 ? 3 : (hx2 ? 4 : fmt_mlp(c)), p.hn_scale = w.hn_scale, p.top_idx = w.top_idx, p.top_val = w.top_val, p.hist = w.hist, p.uimp = w.uimp, p.forced_idx = forced;
  MDM_TRY(moe_route(x, c.M, D, E, p, w.goff, w.cursor, w.perm, w.rowscale, w.pos4, c.s));
  if (route_out && hipMemcpyAsync(route_out, w.top_idx, 4 * c.M * sizeof(int32_t), hipMemcpyDeviceToDevice, c.s) != hipSuccess)
    return MDM_ERR_LAUNCH;
  MdmMlpDesc f = {};
  f.X = (const uint16_t*)w.hn, f.ldx = D, f.gather = w.perm;
  f.M = (int)(4 * c.M), f.Din = D, f.F = F, f.Dout = D;
  f.goff = w.goff, f.ngroups = 2 * E;
  f.w1 = l.w1.hi, f.ldw1 = l.w1.ld, f.w1_gs = (int64_t)F * l.w1.ld, f.b1 = l.b1, f.b1_gs = F;
  f.w2 = l.w2.hi, f.ldw2 = l.w2.ld, f.w2_gs = (int64_t)D * l.w2.ld, f.b2 = l.b2, f.b2_gs = D;
  f.rowscale = w.rowscale, f.r1_scale = 1.f;
  f.C = w.y2, f.ldc = D;
  f.h16 = c.h16;
  f.wstream = l.wstream, f.wstream_gs = l.wstream_gs;
  const bool h = c.bf || c.mix;  // 16-bit expert operands
  if (c.fp8) {
    // fp8 experts (switch_moe.py:19-25,104-109 on e4m3 operands): hidden = e4m3(8 * GELU(dequant(X8 W1_8^T) + b1)), then
    // y2 = prob * (dequant(hidden8 W2_8^T) / 8 + b2).  Row scales come from the router kernel, channel scales from packing.
    const float HS = 8.f;  // static hidden scale: GELU outputs of O(1) land in e4m3's normal range [2^-6, 448]
    {
      GemmArgs g8 = gd(c);
      g8.A.p = w.hn, g8.A.ld = D, g8.A.kind = MDM_OP_FP8_ROW, g8.A.gather = w.perm;
      g8.W.p = l.w1.hi, g8.W.ld = l.w1.ld, g8.W.kind = MDM_OP_FP8_ROW, g8.W.bs1 = (int64_t)F * l.w1.ld;
      g8.a_scale = w.hn_scale, g8.w_scale = (const float*)l.w1.lo;
      g8.goff = w.goff, g8.ngroups = 2 * E;
      g8.M = (int)(4 * c.M), g8.N = F, g8.K = D;
      g8.bias = l.b1, g8.bias_bs = F, g8.act = ACT_GELU;
      g8.C8 = (uint8_t*)w.hid, g8.c8_scale = HS, g8.ldc = F;
      MDM_TRY(gemm(g8, c.s));
    }
    {
      GemmArgs g8 = gd(c);
      g8.A.p = w.hid, g8.A.ld = F, g8.A.kind = MDM_OP_FP8_ROW;
      g8.W.p = l.w2.hi, g8.W.ld = l.w2.ld, g8.W.kind = MDM_OP_FP8_ROW, g8.W.bs1 = (int64_t)D * l.w2.ld;
      g8.a_scale_u = 1.f / HS, g8.w_scale = (const float*)l.w2.lo;
      g8.goff = w.goff, g8.ngroups = 2 * E;
      g8.M = (int)(4 * c.M), g8.N = D, g8.K = F;
      g8.bias = l.b2, g8.bias_bs = D;
      g8.rowscale = w.rowscale;
      g8.C16 = (uint16_t*)w.y2, g8.ldc = D;
      MDM_TRY(gemm(g8, c.s));
    }
    return style_apply(c, l.ffn_style, w.y2, nullptr, nullptr, w.pos4, sc, w.t2, x, 1.f, nullptr, out, out16, true);
  }
  if (h && g_bf16_variant != 21 && fused_mlp_supported(f)) {  // variant 21: two-GEMM chain, for A/B runs
    // throughput mode: both expert GEMMs in one kernel, hidden activations stay in LDS (switch_moe.py:19-25,104-109)
    const bool y16 = c.bf && g_bf16_variant != 25;  // expert outputs stored in 16 bits (what autocast does to a Linear); knob 25 / mixed mode: fp32
    if (y16) f.C = nullptr, f.C16 = (uint16_t*)w.y2;
    const bool pr = g_probe.on && g_probe.n < PROBE_MAX;
    if (pr && hipEventRecord(g_probe.a[g_probe.n], c.s) != hipSuccess) return MDM_ERR_LAUNCH;
    MDM_TRY(fused_mlp(f, c.s));
    if (pr) {
      if (hipEventRecord(g_probe.b[g_probe.n], c.s) != hipSuccess) return MDM_ERR_LAUNCH;
      g_probe.rows[g_probe.n++] = f.M;
    }
    return style_apply(c, l.ffn_style, w.y2, nullptr, nullptr, w.pos4, sc, w.t2, x, 1.f, nullptr, out, out16, y16);
  }
  // measurement probe of the fp32-grade modes: the expert MLP as its two grouped GEMMs, bracketed like the fused kernel above
  const bool pr2 = g_probe.on && g_probe.n < PROBE_MAX;
  if (pr2 && hipEventRecord(g_probe.a[g_probe.n], c.s) != hipSuccess) return MDM_ERR_LAUNCH;
  {
    GemmArgs g = gd(c);  // hidden = GELU(LN_b(x)[routed rows] W1_e^T + b1_e)
    if (h) {
      g.A.p = w.hn, g.A.ld = D, g.A.kind = OP_BF16_ROW, g.precision = 1;
    } else {
      g.A = op_f32(w.hn, D);
      if (hx2) g.A.kind = OP_X2_ROW;
    }
    g.A.gather = w.perm;
    g.W = packed(l.w1);
    g.W.bs1 = (int64_t)F * l.w1.ld;
    g.goff = w.goff, g.ngroups = 2 * E;
    g.M = (int)(4 * c.M), g.N = F, g.K = D;
    g.bias = l.b1, g.bias_bs = F;
    g.act = ACT_GELU;
    g.C = (h || hx2) ? nullptr : w.hid, g.C16 = h ? (uint16_t*)w.hid : nullptr, g.ldc = F;
    if (hx2) g.Cx2 = (uint16_t*)w.hid;
    if (hx2 && l.w1.ws) g.w_stream = l.w1.ws, g.w_stream_gs = gemm_stream3x_group_elems(F, D);  // streamed-weight bf16x3 kernel (csrc/gemm_stream3.hip)
    MDM_TRY(gemm(g, c.s));
  }
  {
    GemmArgs g = gd(c);  // y2[pos] = prob[pos] * (hidden W2_e^T + b2_e)      (switch_moe.py:108-109)
    if (h) {
      g.A.p = w.hid, g.A.ld = F, g.A.kind = OP_BF16_ROW, g.precision = 1;
    } else {
      g.A = op_f32(w.hid, F);
      if (hx2) g.A.kind = OP_X2_ROW;
    }
    g.W = packed(l.w2);
    g.W.bs1 = (int64_t)D * l.w2.ld;
    g.goff = w.goff, g.ngroups = 2 * E;
    g.M = (int)(4 * c.M), g.N = D, g.K = F;
    g.bias = l.b2, g.bias_bs = D;
    g.rowscale = w.rowscale;
    g.C = w.y2, g.ldc = D;  // fp32: the four routed rows of a token are summed in the stylization kernel
    if (hx2 && l.w2.ws) g.w_stream = l.w2.ws, g.w_stream_gs = gemm_stream3x_group_elems(D, F);
    MDM_TRY(gemm(g, c.s));
  }
  if (pr2) {
    if (hipEventRecord(g_probe.b[g_probe.n], c.s) != hipSuccess) return MDM_ERR_LAUNCH;
    g_probe.rows[g_probe.n++] = f.M;
  }
  // mean of the two branches (each the sum of its two routed rows), stylization, residual
  return style_apply(c, l.ffn_style, w.y2, nullptr, nullptr, w.pos4, sc, w.t2, x, 1.f, nullptr, out, out16, false);
}

// Passes the folded text cross-attention (csrc/sdfold.hip) takes, 0 = use the GEMM chain.  Measured end to end (configs[1],
// ms per step, folded / chain): N = 28: 6.09 / 6.28, N = 40: 6.16 / 6.27, N = 64: 6.21 / 6.29 (two passes), N = 85: 6.40 / 6.31
// (four passes: each pass re-streams the x tile and restarts the K' pipeline) -- so the fold is taken up to two passes; knob 24
// forces it at any supported N (tests).
int sd_fold_policy(int D, int H, int N) {
  if (!sd_fold_supported(D, H, N)) return 0;
  const int np = sd_fold_passes(H, N);
  return (np <= 2 || g_bf16_variant == 24) ? np : 0;
}

// MemoryEfficientCrossAttentionBlock (fast_attention.py:301-330); out must not alias x
struct SdFold {  // folded text side of one layer (MdmTextCache.sd_kfold / sd_cb / sd_vfold), or nulls
  const uint16_t* kfold = nullptr;
  const float* cb = nullptr;
  const uint16_t* vfold = nullptr;
};

int sdcross_block(const Ctx& c, const MdmLayer& l, const float* kc, const float* vc, const float* x, const uint16_t* x16,
                  float* out, uint16_t* out16, const SdFold& fold = SdFold()) {
  const MdmModel& m = *c.m;
  const int D = m.D, H = m.H, dh = D / H, N = c.N;
  const Work& w = c.w;
  if (c.bf && fold.kfold && g_bf16_variant != 22 && sd_fold_policy(D, H, N) > 0) {
    // throughput mode: query GEMM + attention core + output GEMM + LayerNorm in one launch (csrc/sdfold.hip)
    MDM_TRY(sd_fold(x16, fold.kfold, fold.cb, fold.vfold, l.sd_out_b, l.sd_ln_w, l.sd_ln_b, c.B, c.S, D, H, N, w.t3,
                    (uint16_t*)w.t4, c.h16, c.s));
    // the 4x FFN pair in one launch (knob 33: two GEMMs) at the full time scale only: at the half scale of the bench batch
    // (6272 rows -> 32-row tiles, every workgroup streams the pair's 4 MB for 32 rows) it measures slower than the two GEMMs
    // (78 vs 66 us; 100 vs 120 us at 12544 rows).  The choice is made on the FRAME count of the scale, never on the batch: a
    // sample's result must not depend on the batch it travels in (shard invariance, cond | uncond batching)
    if (l.sd_ffn_ws && g_bf16_variant != 33 && c.S >= 128) {
      MdmMlpDesc f = {};
      f.X = (const uint16_t*)w.t4, f.ldx = D, f.M = (int)c.M, f.Din = D, f.F = 4 * D, f.Dout = D;
      f.b1 = l.sd_f1_b, f.b2 = l.sd_f2_b, f.wstream = l.sd_ffn_ws, f.wstream_gs = 8 * (int64_t)D * D;
      f.R1 = x, f.ldr1 = D, f.r1_scale = 1.f, f.R2 = w.t3, f.ldr2 = D;  // x + (o + ffn(o))
      f.C = out, f.C16 = out16, f.ldc = D, f.h16 = c.h16;
      if (fused_mlp_stream_supported(f)) return fused_mlp_stream(f, c.s);
    }
    LinOpts o1;
    o1.act = ACT_GELU;
    MDM_TRY(linear_to_act(c, act_of(c, w.t4), c.M, D, l.sd_f1, l.sd_f1_b, 4 * D, w.f1, o1));
    LinOpts o;  // x + (o + ffn(o))
    o.R1 = x, o.R2 = w.t3;
    return linear(c, act_of(c, w.f1), c.M, 4 * D, l.sd_f2, l.sd_f2_b, D, out, out16, o);
  }
  // fp32-grade modes, head_dim 128: the query projection leaves as bf16 hi | lo planes, scores / softmax / PV are one launch on
  // bf16x3 MFMAs (csrc/xattn3.hip).  Knob 56: the chain (two batched contractions around a row softmax).
  const bool fused3 = !c.bf && c.prec == 3 && g_bf16_variant != 56 && xattn3_supported(dh, N) && l.sd_q.lo && D % 32 == 0 &&
                      (c.M * D) % 8 == 0;
  if (fused3) {
    uint16_t* const qh = (uint16_t*)w.t1;
    uint16_t* const ql = qh + c.M * D;
    GemmArgs g = gd(c);
    g.A = op_f32(x, D);
    g.W = packed(l.sd_q);
    g.M = (int)c.M, g.N = D, g.K = D;
    g.bias = l.sd_q_b, g.alpha = 1.f / sqrtf((float)dh);
    g.C16 = qh, g.C16_lo = ql, g.ldc = D;
    MDM_TRY(gemm(g, c.s));
    MDM_TRY(sd_attn3(qh, ql, kc, vc, c.ntok, c.B, c.S, H, dh, N, w.t2, c.x2 ? 1 : 0, c.s));
  } else {
    LinOpts o;
    o.alpha = 1.f / sqrtf((float)dh);
    const bool fz = c.bf && xattn_supported(dh, N) && !(dh == 256 && g_bf16_variant == 23);
    MDM_TRY(linear(c, c.bf ? act_bf16(x16) : act_f32(x), c.M, D, l.sd_q, l.sd_q_b, D, fz ? nullptr : w.t1,
                   fz ? (uint16_t*)w.t1 : nullptr, o));
  }
  if (fused3) {
  } else if (c.bf && xattn_supported(dh, N) && !(dh == 256 && g_bf16_variant == 23)) {
    MDM_TRY(sd_attn(w.t1, c.h16, kc, vc, c.B, c.S, H, dh, N, (uint16_t*)w.t2, nullptr, c.h16, c.s, c.ntok));  // scores, softmax, PV fused
  } else {
    {
      GemmArgs g = gd(c);  // scores[b,h,s,n] = q . k
      g.A = op_f32(w.t1, D);
      g.A.bs1 = (int64_t)c.S * D, g.A.bs2 = dh;
      g.W = op_f32(kc, D);
      g.W.bs1 = (int64_t)N * D, g.W.bs2 = dh;
      g.M = c.S, g.N = N, g.K = dh;
      g.batch = c.B * H, g.nb2 = H;
      g.C = w.scr, g.ldc = N, g.c_bs1 = (int64_t)H * c.S * N, g.c_bs2 = (int64_t)c.S * N;
      MDM_TRY(gemm(g, c.s));
    }
    MDM_TRY(row_softmax(w.scr, c.M * H, N, c.s, c.ntok, (int64_t)H * c.S));
    {
      GemmArgs g = gd(c);  // o[b,s,h,:] = p v
      g.A = op_f32(w.scr, N);
      g.A.bs1 = (int64_t)H * c.S * N, g.A.bs2 = (int64_t)c.S * N;
      g.W = op_f32_kstride(vc, D);
      g.W.bs1 = (int64_t)N * D, g.W.bs2 = dh;
      g.M = c.S, g.N = dh, g.K = N;
      g.batch = c.B * H, g.nb2 = H;
      g.C = c.bf ? nullptr : w.t2, g.C16 = c.bf ? (uint16_t*)w.t2 : nullptr;
      g.ldc = D, g.c_bs1 = (int64_t)c.S * D, g.c_bs2 = dh;
      MDM_TRY(gemm(g, c.s));
    }
  }
  MDM_TRY(linear(c, Act{w.t2, c.bf, fused3 && c.x2}, c.M, D, l.sd_out, l.sd_out_b, D, w.t3, nullptr));
  // the 4x FFN (LayerNorm -> Linear -> GELU -> Linear): 16-bit operands in the throughput AND the mixed mode; in the fp32-grade
  // mode the LayerNorm rows and the hidden rows are written pre-split for the GEMM that reads them
  const bool h = c.bf || c.mix, fx2 = !h && c.x2;
  MDM_TRY(ln_chain(w.t3, c.M, D, l.sd_ln_w, l.sd_ln_b, w.t4, fx2 ? 4 : fmt_mlp(c), nullptr, nullptr, nullptr, 0, c.s));
  {
    LinOpts o;
    o.act = ACT_GELU;
    if (fx2) o.outx2 = (uint16_t*)w.f1;
    MDM_TRY(linear(c, Act{w.t4, h, fx2}, c.M, D, l.sd_f1, l.sd_f1_b, 4 * D, (h || fx2) ? nullptr : w.f1, h ? (uint16_t*)w.f1 : nullptr, o));
  }
  LinOpts o;  // x + (o + ffn(o))
  o.R1 = x, o.R2 = w.t3;
  return linear(c, Act{w.f1, h, fx2}, c.M, 4 * D, l.sd_f2, l.sd_f2_b, D, out, out16, o);
}

const float* tc_at(const MdmModel& m, const MdmTextCache& tc, int layer) {
  const int dh = m.D / m.H;
  return tc.lin_at + (int64_t)layer * tc.B * m.H * dh * dh;
}
const float* tc_k(const MdmModel& m, const MdmTextCache& tc, int layer) {
  return tc.sd_k + (int64_t)layer * tc.B * tc.N * m.D;
}
const float* tc_v(const MdmModel& m, const MdmTextCache& tc, int layer) {
  return tc.sd_v + (int64_t)layer * tc.B * tc.N * m.D;
}
SdFold tc_fold(const MdmModel& m, const MdmTextCache& tc, int layer) {
  SdFold f;
  if (tc.sd_kfold && tc.sd_cb && tc.sd_vfold) {  // per layer: [B][passes][128][D], [B][passes][128], [B][passes][D][128]
    const int64_t np = sd_fold_passes(m.H, tc.N);
    f.kfold = tc.sd_kfold + (int64_t)layer * tc.B * np * 128 * m.D;
    f.cb = tc.sd_cb + (int64_t)layer * tc.B * np * 128;
    f.vfold = tc.sd_vfold + (int64_t)layer * tc.B * np * m.D * 128;
  }
  return f;
}

// one MoEExtendedDecoderLayer (transformer.py:55-64): x (+ bf16 shadow x16) is updated in place, (y, y16) is scratch
int decoder_layer(const Ctx& c, int layer, const MdmTextCache& tc, float* x, uint16_t* x16, float* y, uint16_t* y16,
                  const float* sc4, const int* forced, float* trace, int32_t* route_out = nullptr) {
  const MdmModel& m = *c.m;
  const MdmLayer& l = m.layers[layer];
  const int64_t scs = (int64_t)c.B * 2 * m.D, n = c.M * m.D;
  auto dump = [&](int slot, const float* p) -> int {
    if (!trace) return MDM_OK;
    return hipMemcpyAsync(trace + (int64_t)slot * n, p, n * sizeof(float), hipMemcpyDeviceToDevice, c.s) == hipSuccess
               ? MDM_OK
               : MDM_ERR_LAUNCH;
  };
  if (!c.bf) x16 = y16 = nullptr;  // the 16-bit shadows of the residual stream are read by the 16-bit modes only: do not write them here
  MDM_TRY(dual_block(c, l, x, x16, sc4, y, l.ca_norm_w, l.ca_norm_b));
  MDM_TRY(dump(0, y));
  MDM_TRY(cross_block(c, l, tc_at(m, tc, layer), y, sc4 + 2 * scs, x, true));
  MDM_TRY(dump(1, x));
  MDM_TRY(moe_block(c, l, x, sc4 + 3 * scs, forced, y, y16, route_out));
  MDM_TRY(dump(2, y));
  MDM_TRY(sdcross_block(c, l, tc_k(m, tc, layer), tc_v(m, tc, layer), y, y16, x, x16, tc_fold(m, tc, layer)));
  return dump(3, x);
}

// fused time/text embedding + every stylization block's (scale|shift)  (transformer.py:313-321, stylization.py:22-27)
// gt = gated_fusion.proj_time(time_proj(time_embed(learnable_time_embed(t))))  [B, D] -> dst   (transformer.py:318-320)
int stem_time_branch(const Ctx& c, const int64_t* timesteps, int B, float* dst) {
  const MdmModel& m = *c.m;
  const int D = m.D, Te = 4 * D;
  const Work& w = c.w;
  LinOpts silu_o;
  silu_o.act = ACT_SILU;
  MDM_TRY(sinusoid(timesteps, B, D, w.s_a, c.s));
  MDM_TRY(linear(c, act_f32(w.s_a), B, D, m.tmlp0, m.tmlp0_b, 2 * D, w.s_b, nullptr, silu_o));
  MDM_TRY(linear(c, act_f32(w.s_b), B, 2 * D, m.tmlp2, m.tmlp2_b, D, w.s_a, nullptr));
  MDM_TRY(linear(c, act_f32(w.s_a), B, D, m.te0, m.te0_b, Te, w.s_b, nullptr, silu_o));
  MDM_TRY(linear(c, act_f32(w.s_b), B, Te, m.te2, m.te2_b, Te, w.s_c, nullptr));
  MDM_TRY(linear(c, act_f32(w.s_c), B, Te, m.tproj, m.tproj_b, D, w.s_a, nullptr));
  return linear(c, act_f32(w.s_a), B, D, m.gf_time, m.gf_time_b, D, dst, nullptr);
}

// gx = gated_fusion.proj_text([text_proj](xf_proj))  [B, D] -> dst      (transformer.py:313-315, gate.py:17)
int stem_text_branch(const Ctx& c, const float* xf_proj, int B, float* dst) {
  const MdmModel& m = *c.m;
  const float* tp = xf_proj;
  if (m.Dt != m.D) {  // the per-call random text_proj, captured
    if (!m.text_proj.hi) return MDM_ERR_ARG;
    MDM_TRY(linear(c, act_f32(xf_proj), B, m.Dt, m.text_proj, m.text_proj_b, m.D, c.w.s_c, nullptr));
    tp = c.w.s_c;
  }
  return linear(c, act_f32(tp), B, m.D, m.gf_text, m.gf_text_b, m.D, dst, nullptr);
}

// fused time/text embedding + every stylization block's (scale|shift)  (transformer.py:313-321, stylization.py:22-27)
int stem_embeddings(const Ctx& c, const int64_t* timesteps, const float* xf_proj, const MdmStemCache* sc_cache,
                    float* emb_out, float* sc_out) {
  const MdmModel& m = *c.m;
  const int D = m.D, Te = 4 * D, B = c.B, nblk = 8 * m.L;
  const Work& w = c.w;
  LinOpts silu_o;
  silu_o.act = ACT_SILU;
  // fused = sigmoid(t + x) * t + (1 - sigmoid) * x                  (gate.py:18-20) -> mix (fp32, or bf16 in throughput mode)
  float* mix = w.s_b;
  if (sc_cache && sc_cache->time_table && sc_cache->gx) {
    MDM_TRY(gated_mix_gather(sc_cache->time_table, timesteps, sc_cache->steps, sc_cache->gx, B, D, c.bf ? nullptr : mix,
                             c.bf ? (uint16_t*)mix : nullptr, c.h16, c.s));
  } else {
    MDM_TRY(stem_time_branch(c, timesteps, B, w.s_b));
    MDM_TRY(stem_text_branch(c, xf_proj, B, w.s_a));
    MDM_TRY(gated_mix(w.s_b, w.s_a, (int64_t)B * D, w.s_c, c.s));
    if (c.bf) {
      MDM_TRY(to_bf16(w.s_c, (int64_t)B * D, (uint16_t*)w.s_b, c.h16, c.s));
    } else {
      mix = w.s_c;
    }
  }
  MDM_TRY(linear_to_act(c, act_of(c, mix), B, D, m.gf_post0, m.gf_post0_b, D, w.s_a, silu_o));
  float* emb = emb_out ? emb_out : w.emb;
  MDM_TRY(linear(c, act_of(c, w.s_a), B, D, m.gf_post2, m.gf_post2_b, D, emb, c.bf ? (uint16_t*)w.s_c : nullptr));
  // all 8L blocks at once: e1 = SiLU(emb Weph^T + beph) [B, 8L*Te]; sc[j] = e1[:, j] W1_j^T + b1_j [8L, B, 2D]
  MDM_TRY(linear_to_act(c, c.bf ? act_bf16((uint16_t*)w.s_c) : act_f32(emb), B, D, m.style_eph, m.style_eph_b, nblk * Te,
                        w.e1, silu_o));
  GemmArgs g = gd(c);
  if (c.bf) {
    g.A.p = w.e1, g.A.ld = (int64_t)nblk * Te, g.A.kind = OP_BF16_ROW, g.precision = 1;
  } else {
    g.A = op_f32(w.e1, (int64_t)nblk * Te);
  }
  g.A.bs1 = Te;
  g.W = packed(m.style_emb);
  g.W.bs1 = (int64_t)2 * D * m.style_emb.ld;
  g.M = B, g.N = 2 * D, g.K = Te;
  g.batch = nblk, g.nb2 = 1;
  g.bias = m.style_emb_b, g.bias_bs = 2 * D;
  g.C = sc_out, g.ldc = 2 * D, g.c_bs1 = (int64_t)B * 2 * D;
  return gemm(g, c.s);
}

int check_model(const MdmModel* m) {
  if (!m || !m->layers || m->D <= 0 || m->H <= 0 || m->D % m->H || m->L <= 0 || m->E < 2 || m->E > 16) return MDM_ERR_ARG;
  const int dh = m->D / m->H;
  if (dh != 16 && dh != 32 && dh != 64 && dh != 128 && dh != 256) return MDM_ERR_UNSUPPORTED;
  if (m->D > 1024 || m->Dt > 1024 || (m->D & 3) || (m->F & 3)) return MDM_ERR_UNSUPPORTED;
  return MDM_OK;
}

}  // namespace
}  // namespace mdm

using namespace mdm;

extern "C" {

int64_t mdm_workspace_bytes(const MdmModel* m, int32_t B, int32_t T, int32_t N) {
  if (check_model(m) != MDM_OK || B <= 0 || T <= 0) return -1;
  return carve(*m, B, T, N, nullptr).bytes;
}

int mdm_text_cache_build(const MdmModel* m, const float* xf_out, const MdmTextCache* tc, void* ws, int64_t ws_bytes,
                         int32_t precision, void* stream) {
  MDM_TRY(check_model(m));
  if (!xf_out || !tc || !tc->lin_at || !tc->sd_k || !tc->sd_v || tc->B <= 0 || tc->N <= 0 || tc->N > 128 || !ws)
    return MDM_ERR_ARG;
  Ctx c = {};
  // computed once per caption batch, never per step: always the bf16x3 arithmetic (its weights are packed bf16 hi + lo in
  // every mode); `precision` only selects the 16-bit format of the folded K' / V' images
  c.m = m, c.s = (hipStream_t)stream, c.prec = 3, c.bf = false, c.mix = false, c.B = tc->B, c.N = tc->N;
  c.ntok = tc->ntok;
  c.h16 = (precision == MDM_PREC_F16 || precision == MDM_PREC_MIXED || precision == MDM_PREC_FP8) ? MDM_H16_F16 : MDM_H16_BF16;
  c.w = carve(*m, tc->B, 2, tc->N, ws);
  if (c.w.bytes > ws_bytes) return MDM_ERR_ARG;
  const int D = m->D, H = m->H, dh = D / H, N = tc->N, B = tc->B;
  const int64_t BN = (int64_t)B * N;
  for (int layer = 0; layer < 2 * m->L; ++layer) {
    const MdmLayer& l = m->layers[layer];
    MDM_TRY(ln_chain(xf_out, BN, m->Dt, l.ca_tnorm_w, l.ca_tnorm_b, c.w.tn, 0, nullptr, nullptr, nullptr, 0, c.s));
    MDM_TRY(linear(c, act_f32(c.w.tn), BN, m->Dt, l.ca_k, l.ca_k_b, D, c.w.kb, nullptr));
    MDM_TRY(col_softmax(c.w.kb, B, N, D, c.s, tc->ntok));  // softmax over text tokens (fast_attention.py:249)
    MDM_TRY(linear(c, act_f32(c.w.tn), BN, m->Dt, l.ca_v, l.ca_v_b, D, c.w.vb, nullptr));
    GemmArgs g = gemm_defaults(3);  // A^T[b,h][l][d] = sum_n v[n,l] k[n,d]   (fast_attention.py:252)
    g.A = op_f32_kstride(c.w.vb, D);
    g.A.bs1 = (int64_t)N * D, g.A.bs2 = dh;
    g.W = op_f32_kstride(c.w.kb, D);
    g.W.bs1 = (int64_t)N * D, g.W.bs2 = dh;
    g.M = dh, g.N = dh, g.K = N;
    g.batch = B * H, g.nb2 = H;
    g.C = (float*)tc_at(*m, *tc, layer), g.ldc = dh, g.c_bs1 = (int64_t)H * dh * dh, g.c_bs2 = (int64_t)dh * dh;
    MDM_TRY(gemm(g, c.s));
    MDM_TRY(linear(c, act_f32(xf_out), BN, m->Dt, l.sd_k, l.sd_k_b, D, (float*)tc_k(*m, *tc, layer), nullptr));
    MDM_TRY(linear(c, act_f32(xf_out), BN, m->Dt, l.sd_v, l.sd_v_b, D, (float*)tc_v(*m, *tc, layer), nullptr));
    if (tc->sd_kfold && tc->sd_cb && tc->sd_vfold) {
      // fold the query / output projections into the text side (csrc/sdfold.hip), fp32-grade arithmetic, bf16 results
      if (!l.sd_q_w32 || !l.sd_out_w32 || sd_fold_policy(D, H, N) <= 0) return MDM_ERR_ARG;
      const SdFold f = tc_fold(*m, *tc, layer);
      const float scale = 1.f / sqrtf((float)dh);
      const float* kc = tc_k(*m, *tc, layer);
      const float* vc = tc_v(*m, *tc, layer);
      const int hpp = sd_fold_heads_per_pass(H, N), np = sd_fold_passes(H, N);
      for (int ps = 0; ps < np; ++ps) {  // one pass = hc whole heads, packed as rows / columns hs * N + n of a 128-wide image
        const int h0 = ps * hpp, hc = (H - h0) < hpp ? (H - h0) : hpp;
        {  // K'[b][ps][hs*N + n][j] = scale * sum_d key[b,n,h*dh+d] Wq[h*dh+d, j]
          GemmArgs g = gemm_defaults(3);
          g.A = op_f32(kc + h0 * dh, D);
          g.A.bs1 = (int64_t)N * D, g.A.bs2 = dh;
          g.W = op_f32_kstride(l.sd_q_w32 + (int64_t)h0 * dh * D, D);
          g.W.bs1 = 0, g.W.bs2 = (int64_t)dh * D;
          g.M = N, g.N = D, g.K = dh;
          g.batch = B * hc, g.nb2 = hc;
          g.alpha = scale;
          g.h16 = c.h16;
          g.C16 = (uint16_t*)f.kfold + (int64_t)ps * 128 * D, g.ldc = D, g.c_bs1 = (int64_t)np * 128 * D, g.c_bs2 = (int64_t)N * D;
          MDM_TRY(gemm(g, c.s));
        }
        {  // cb[b][ps][hs*N + n] = scale * key[b,n,h*dh:] . bq[h*dh:]
          GemmArgs g = gemm_defaults(3);
          g.A = op_f32(kc + h0 * dh, D);
          g.A.bs1 = (int64_t)N * D, g.A.bs2 = dh;
          g.W = op_f32(l.sd_q_b + h0 * dh, dh);
          g.W.bs1 = 0, g.W.bs2 = dh;
          g.M = N, g.N = 1, g.K = dh;
          g.batch = B * hc, g.nb2 = hc;
          g.alpha = scale;
          g.C = (float*)f.cb + ps * 128, g.ldc = 1, g.c_bs1 = (int64_t)np * 128, g.c_bs2 = N;
          MDM_TRY(gemm(g, c.s));
        }
        {  // V'^T[b][ps][j][hs*N + n] = sum_d Wout[j, h*dh+d] value[b,n,h*dh+d]
          GemmArgs g = gemm_defaults(3);
          g.A = op_f32(l.sd_out_w32 + h0 * dh, D);
          g.A.bs1 = 0, g.A.bs2 = dh;
          g.W = op_f32(vc + h0 * dh, D);
          g.W.bs1 = (int64_t)N * D, g.W.bs2 = dh;
          g.M = D, g.N = N, g.K = dh;
          g.batch = B * hc, g.nb2 = hc;
          g.h16 = c.h16;
          g.C16 = (uint16_t*)f.vfold + (int64_t)ps * D * 128, g.ldc = 128, g.c_bs1 = (int64_t)np * D * 128, g.c_bs2 = N;
          MDM_TRY(gemm(g, c.s));
        }
      }
      // per-sample token counts: the folded columns of a sample's padding tokens get a bias that makes their probability 0
      MDM_TRY(sd_fold_mask_cb((float*)f.cb, B, np, hpp, N, tc->ntok, c.s));
    }
  }
  return MDM_OK;
}

int mdm_stem_embeddings(const MdmModel* m, const int64_t* timesteps, const float* xf_proj, int32_t B, float* emb_out,
                        float* sc_out, void* ws, int64_t ws_bytes, int32_t precision, void* stream) {
  MDM_TRY(check_model(m));
  if (!timesteps || !xf_proj || !sc_out || !ws || B <= 0) return MDM_ERR_ARG;
  Ctx c = {};
  c.m = m, c.s = (hipStream_t)stream, c.B = B;
  if (!set_precision(c, m, precision)) return MDM_ERR_UNSUPPORTED;
  c.w = carve(*m, B, 2, 1, ws);
  if (c.w.bytes > ws_bytes) return MDM_ERR_ARG;
  return stem_embeddings(c, timesteps, xf_proj, nullptr, emb_out, sc_out);
}

int mdm_stem_cache_build(const MdmModel* m, int32_t steps, float* time_table, const float* xf_proj, int32_t B, float* gx,
                         void* ws, int64_t ws_bytes, int32_t precision, void* stream) {
  MDM_TRY(check_model(m));
  if (!ws || (time_table && steps <= 0) || (gx && (!xf_proj || B <= 0))) return MDM_ERR_ARG;
  constexpr int CH = 128;
  Ctx c = {};
  // tabulated in the fp32-grade arithmetic regardless of the run mode: it is computed once per loop
  c.m = m, c.s = (hipStream_t)stream, c.prec = 3, c.bf = false, c.mix = false, c.h16 = MDM_H16_BF16, c.B = CH;
  (void)precision;
  c.w = carve(*m, CH, 2, 1, ws);
  if (c.w.bytes > ws_bytes) return MDM_ERR_ARG;
  int64_t* ts = (int64_t*)c.w.hid;  // scratch for the timestep ramp
  if (time_table)
    for (int t0 = 0; t0 < steps; t0 += CH) {
      const int n = steps - t0 < CH ? steps - t0 : CH;
      MDM_TRY(iota_i64(ts, n, t0, c.s));
      MDM_TRY(stem_time_branch(c, ts, n, time_table + (int64_t)t0 * m->D));
    }
  if (gx)
    for (int b0 = 0; b0 < B; b0 += CH) {
      const int n = B - b0 < CH ? B - b0 : CH;
      MDM_TRY(stem_text_branch(c, xf_proj + (int64_t)b0 * m->Dt, n, gx + (int64_t)b0 * m->D));
    }
  return MDM_OK;
}

int mdm_denoiser_forward(const MdmModel* m, const MdmTextCache* tc, const float* x, const int64_t* timesteps,
                         const int32_t* length, const float* xf_proj, int32_t B, int32_t T, float* out, void* ws,
                         int64_t ws_bytes, const int32_t* forced_routing, float* trace, const MdmStemCache* stem,
                         int32_t precision, void* stream) {
  MDM_TRY(check_model(m));
  if (!tc || !x || !timesteps || !length || !xf_proj || !out || !ws || B <= 0 || T <= 0) return MDM_ERR_ARG;
  if (T % 2 || T > m->num_frames) return MDM_ERR_ARG;  // odd T breaks the U-shape (transformer.py:223-224,353)
  if (tc->B != B) return MDM_ERR_ARG;
  Ctx c = {};
  c.m = m, c.s = (hipStream_t)stream, c.B = B, c.N = tc->N;
  c.ntok = tc->ntok;
  if (!set_precision(c, m, precision)) return MDM_ERR_UNSUPPORTED;
  c.w = carve(*m, B, T, tc->N, ws);
  if (c.w.bytes > ws_bytes) return MDM_ERR_ARG;
  const Work& w = c.w;
  const int D = m->D, L = m->L;
  const int64_t Mfull = (int64_t)B * T, Mlow = Mfull / 2;
  const int64_t scl = (int64_t)4 * B * 2 * D;  // (scale|shift) floats per layer
  MDM_TRY(stem_embeddings(c, timesteps, xf_proj, stem, nullptr, w.sc));
  // h = joint_embed(x) + sequence_embedding[:T]                     (transformer.py:324-326)
  {
    LinOpts o;
    o.R1 = m->seq_emb, o.r1_mod = T;
    // the root of the residual stream (and of the U's skip connection): always the bf16x3 arithmetic -- 3.4 GFLOP, K = 263.
    // Measured at B = 32 / T = 196 / L = 4, fp16 mode, free routing: 1679 -> 596 of 75264 routing decisions differ from the
    // fp32-grade run, median frame error 4.7e-3 -> 1.8e-3 (tools/mode_compare.py; knob 31 restores the single bf16 pass).
    // The same treatment of down / up / the stem GEMMs bought another 14 % for +0.2 ms per step: not taken.
    Ctx cj = c;
    if (g_bf16_variant != 31) cj.prec = 3;
    MDM_TRY(linear(cj, act_f32(x), Mfull, m->feats, m->joint, m->joint_b, D, w.h0, c.bf ? w.h016 : nullptr, o));
  }
  // Conv1d(k=2,s=2) == Linear over pairs of frames                  (:332-337)
  MDM_TRY(linear(c, c.bf ? act_bf16(w.h016) : act_f32(w.h0), Mlow, 2 * D, m->down, m->down_b, D, w.xa,
                 c.bf ? w.xa16 : nullptr));
  MDM_TRY(halve_lengths(length, B, w.len_low, c.s));  // (:341-342)
  c.S = T / 2, c.M = Mlow, c.len = w.len_low;
  if (g_route_dump && g_route_cap < (int64_t)2 * L * 4 * Mfull) return MDM_ERR_ARG;  // the dump buffer is too small for this forward
  for (int i = 0; i < L; ++i) {  // coarse scale blocks, in place on xa (xb = scratch)      (:343-344)
    const int32_t* fr = forced_routing ? forced_routing + (int64_t)i * 4 * Mfull : nullptr;
    float* tr = trace ? trace + (int64_t)i * 4 * Mfull * D : nullptr;
    int32_t* rd = g_route_dump ? g_route_dump + (int64_t)i * 4 * Mfull : nullptr;
    MDM_TRY(decoder_layer(c, i, *tc, w.xa, w.xa16, w.xb, w.xb16, w.sc + i * scl, fr, tr, rd));
  }
  // ConvTranspose1d(k=2,s=2) == Linear D -> 2D per coarse frame, rows (B*T/2, 2D) == (B*T, D); + skip h  (:347-353)
  {
    LinOpts o;
    o.R1 = w.h0;
    MDM_TRY(linear(c, c.bf ? act_bf16(w.xa16) : act_f32(w.xa), Mlow, D, m->up, m->up_b2, 2 * D, w.xb,
                   c.bf ? w.xb16 : nullptr, o));
  }
  c.S = T, c.M = Mfull, c.len = length;
  for (int i = 0; i < L; ++i) {  // full scale blocks, in place on xb                        (:356-357)
    const int32_t* fr = forced_routing ? forced_routing + (int64_t)(L + i) * 4 * Mfull : nullptr;
    float* tr = trace ? trace + (int64_t)(L + i) * 4 * Mfull * D : nullptr;
    int32_t* rd = g_route_dump ? g_route_dump + (int64_t)(L + i) * 4 * Mfull : nullptr;
    MDM_TRY(decoder_layer(c, L + i, *tc, w.xb, w.xb16, w.xa, w.xa16, w.sc + (L + i) * scl, fr, tr, rd));
  }
  return linear(c, c.bf ? act_bf16(w.xb16) : act_f32(w.xb), Mfull, D, m->out, m->out_b, m->feats, out, nullptr);  // (:360)
}

int mdm_block_forward(const MdmModel* m, int32_t layer, int32_t block, const MdmTextCache* tc, const float* h,
                      const float* sc, const int32_t* len, int32_t B, int32_t S, float* out, void* ws, int64_t ws_bytes,
                      const int32_t* forced_routing, int32_t precision, void* stream) {
  MDM_TRY(check_model(m));
  if (!h || !sc || !len || !out || !ws || B <= 0 || S <= 0 || layer < 0 || layer >= 2 * m->L) return MDM_ERR_ARG;
  if ((block == MDM_BLOCK_CROSS || block == MDM_BLOCK_SDCROSS || block == MDM_BLOCK_LAYER) && (!tc || tc->B != B))
    return MDM_ERR_ARG;
  Ctx c = {};
  c.m = m, c.s = (hipStream_t)stream, c.B = B, c.S = S, c.M = (int64_t)B * S, c.len = len;
  if (!set_precision(c, m, precision)) return MDM_ERR_UNSUPPORTED;
  c.N = tc ? tc->N : 1;
  c.ntok = tc ? tc->ntok : nullptr;
  c.w = carve(*m, B, S, c.N, ws);
  if (c.w.bytes > ws_bytes) return MDM_ERR_ARG;
  const MdmLayer& l = m->layers[layer];
  const int64_t scs = (int64_t)B * 2 * m->D, n = c.M * m->D;
  if (c.bf && (n & 3)) return MDM_ERR_UNSUPPORTED;
  if (c.bf) MDM_TRY(to_bf16(h, n, c.w.h016, c.h16, c.s));  // callers hand over fp32 only: build the 16-bit shadow here
  switch (block) {
    case MDM_BLOCK_DUAL: return dual_block(c, l, h, c.w.h016, sc, out);
    case MDM_BLOCK_CROSS: return cross_block(c, l, tc_at(*m, *tc, layer), h, sc + 2 * scs, out);
    case MDM_BLOCK_MOE:
      if (g_route_dump && g_route_cap < 4 * c.M) return MDM_ERR_ARG;
      return moe_block(c, l, h, sc + 3 * scs, forced_routing, out, nullptr, g_route_dump);
    case MDM_BLOCK_SDCROSS:
      return sdcross_block(c, l, tc_k(*m, *tc, layer), tc_v(*m, *tc, layer), h, c.w.h016, out, nullptr,
                           tc_fold(*m, *tc, layer));
    case MDM_BLOCK_LAYER: {
      if (hipMemcpyAsync(c.w.xa, h, n * sizeof(float), hipMemcpyDeviceToDevice, c.s) != hipSuccess) return MDM_ERR_LAUNCH;
      if (c.bf) MDM_TRY(to_bf16(h, n, c.w.xa16, c.h16, c.s));
      MDM_TRY(decoder_layer(c, layer, *tc, c.w.xa, c.w.xa16, c.w.xb, c.w.xb16, sc, forced_routing, nullptr));
      return hipMemcpyAsync(out, c.w.xa, n * sizeof(float), hipMemcpyDeviceToDevice, c.s) == hipSuccess ? MDM_OK
                                                                                                         : MDM_ERR_LAUNCH;
    }
    default: return MDM_ERR_ARG;
  }
}

// ---- per-block entry points named as SURVEY.md §8(b) lists them: thin views of mdm_block_forward -----------------------
int mdm_moe_ffn_forward(const MdmModel* m, int32_t layer, const float* h, const float* sc4, const int32_t* len, int32_t B,
                        int32_t S, float* out, void* ws, int64_t ws_bytes, const int32_t* forced_routing,
                        int32_t precision, void* stream) {
  return mdm_block_forward(m, layer, MDM_BLOCK_MOE, nullptr, h, sc4, len, B, S, out, ws, ws_bytes, forced_routing, precision,
                           stream);
}
int mdm_dual_self_attn_forward(const MdmModel* m, int32_t layer, const float* h, const float* sc4, const int32_t* len,
                               int32_t B, int32_t S, float* out, void* ws, int64_t ws_bytes, int32_t precision,
                               void* stream) {
  return mdm_block_forward(m, layer, MDM_BLOCK_DUAL, nullptr, h, sc4, len, B, S, out, ws, ws_bytes, nullptr, precision, stream);
}
int mdm_linear_xattn_forward(const MdmModel* m, int32_t layer, const MdmTextCache* tc, const float* h, const float* sc4,
                             const int32_t* len, int32_t B, int32_t S, float* out, void* ws, int64_t ws_bytes,
                             int32_t precision, void* stream) {
  return mdm_block_forward(m, layer, MDM_BLOCK_CROSS, tc, h, sc4, len, B, S, out, ws, ws_bytes, nullptr, precision, stream);
}
int mdm_softmax_xattn_ffn_forward(const MdmModel* m, int32_t layer, const MdmTextCache* tc, const float* h,
                                  const float* sc4, const int32_t* len, int32_t B, int32_t S, float* out, void* ws,
                                  int64_t ws_bytes, int32_t precision, void* stream) {
  return mdm_block_forward(m, layer, MDM_BLOCK_SDCROSS, tc, h, sc4, len, B, S, out, ws, ws_bytes, nullptr, precision, stream);
}
// one PerformerSelfAttention (fast_attention.py:137-179): which = 0 local_attn, 1 global_attn; sc = that block's
// (scale|shift) rows [B, 2D]
int mdm_performer_attn_forward(const MdmModel* m, int32_t layer, int32_t which, const float* h, const float* sc,
                               const int32_t* len, int32_t B, int32_t S, float* out, void* ws, int64_t ws_bytes,
                               int32_t precision, void* stream) {
  MDM_TRY(check_model(m));
  if (!h || !sc || !len || !out || !ws || B <= 0 || S <= 0 || layer < 0 || layer >= 2 * m->L || which < 0 || which > 1)
    return MDM_ERR_ARG;
  Ctx c = {};
  c.m = m, c.s = (hipStream_t)stream, c.B = B, c.S = S, c.M = (int64_t)B * S, c.len = len, c.N = 1;
  if (!set_precision(c, m, precision)) return MDM_ERR_UNSUPPORTED;
  c.w = carve(*m, B, S, 1, ws);
  if (c.w.bytes > ws_bytes) return MDM_ERR_ARG;
  const MdmLayer& l = m->layers[layer];
  const MdmPerformer& p = which ? l.global : l.local;
  MDM_TRY(ln_chain(h, c.M, m->D, p.pre_w, p.pre_b, c.w.t3, fmt16(c), nullptr, nullptr, nullptr, 0, c.s));
  return performer(c, p, h, act_of(c, c.w.t3), sc, out);
}

int mdm_stylization_forward(const MdmStyle* st, const float* h, const float* sc, int32_t B, int32_t S, int32_t D,
                            float* tmp, float* out, int32_t precision, void* stream) {
  if (!st || !h || !sc || !tmp || !out || B <= 0 || S <= 0) return MDM_ERR_ARG;
  MdmModel fake = {};
  fake.D = D;
  Ctx c = {};
  fake.F = D;
  c.m = &fake, c.s = (hipStream_t)stream, c.B = B, c.S = S, c.M = (int64_t)B * S;
  if (!set_precision(c, &fake, precision)) return MDM_ERR_UNSUPPORTED;
  return style_apply(c, *st, h, nullptr, nullptr, nullptr, sc, tmp, nullptr, 1.f, nullptr, out);
}

int64_t mdm_text_head_workspace_bytes(int32_t B, int32_t N0, int32_t P, int32_t Hs, int32_t Dt) {
  if (B <= 0 || N0 < 0 || P < 0 || Hs <= 0 || Dt <= 0) return -1;
  const int64_t fl = (int64_t)B * N0 * Hs + (int64_t)P * Hs + (int64_t)B * N0 * Dt + (int64_t)P * Dt;
  return ((fl * 4 + 255) & ~(int64_t)255) + 1024;
}

// EnhancedTextEncoder's projection head (text_encoder.py:13-18,39-43) on the hidden states of any text encoder
int mdm_text_head_forward(const float* hidden, const float* prompts, const float* ln_w, const float* ln_b,
                          const MdmPacked* w, const float* bias, int32_t B, int32_t N0, int32_t P, int32_t Hs,
                          int32_t Dt, float* xf_out, float* xf_proj, void* ws, int64_t ws_bytes, int32_t precision,
                          void* stream) {
  if (!ln_w || !ln_b || !w || !w->hi || !xf_out || !xf_proj || !ws || B <= 0 || N0 < 0 || P < 0 || N0 + P <= 0 ||
      (N0 > 0 && !hidden) || (P > 0 && !prompts))
    return MDM_ERR_ARG;
  if (precision != 1 && precision != 3) return MDM_ERR_ARG;
  if (precision == 3 && !w->lo) return MDM_ERR_ARG;
  if (ws_bytes < mdm_text_head_workspace_bytes(B, N0, P, Hs, Dt)) return MDM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  Bump b(ws);
  float* lnh = b.take<float>((int64_t)B * N0 * Hs);
  float* lnp = b.take<float>((int64_t)P * Hs);
  float* ph = b.take<float>((int64_t)B * N0 * Dt);
  float* pp = b.take<float>((int64_t)P * Dt);
  MdmModel fake = {};
  Ctx c = {};
  c.m = &fake, c.s = s, c.prec = precision, c.bf = false, c.mix = false, c.h16 = MDM_H16_BF16;
  LinOpts o;
  o.act = ACT_GELU;
  if (N0 > 0) {
    MDM_TRY(ln_chain(hidden, (int64_t)B * N0, Hs, ln_w, ln_b, lnh, 0, nullptr, nullptr, nullptr, 0, s));
    MDM_TRY(linear(c, act_f32(lnh), (int64_t)B * N0, Hs, *w, bias, Dt, ph, nullptr, o));
  }
  if (P > 0) {  // the prompt rows are the same for every caption: projected once, broadcast by the assemble kernel
    MDM_TRY(ln_chain(prompts, P, Hs, ln_w, ln_b, lnp, 0, nullptr, nullptr, nullptr, 0, s));
    MDM_TRY(linear(c, act_f32(lnp), P, Hs, *w, bias, Dt, pp, nullptr, o));
  }
  return text_assemble(pp, ph, B, N0, P, Dt, xf_out, xf_proj, s);
}

// generated motions -> joint positions (tools/visualization.py:21-27,89; utils/motion_process.py:362-416; utils/utils.py:125-130)
int mdm_motion_postprocess(const float* motion, const int32_t* length, const float* mean, const float* std, int32_t B,
                           int32_t T, int32_t feats, int32_t joints, int32_t radius, const double* weights,
                           float* scratch, float* joints_out, void* stream) {
  if (!motion || !mean || !std || !scratch || !joints_out || B <= 0 || T <= 0 || joints < 1 || radius < 0 ||
      (radius > 0 && !weights))
    return MDM_ERR_ARG;
  if (feats < 4 + 3 * (joints - 1)) return MDM_ERR_ARG;
  return motion_post(motion, length, mean, std, B, T, feats, joints, radius, weights, scratch, joints_out,
                     (hipStream_t)stream);
}

// passes of <= 128 folded text columns that sd_fold takes for (H heads, N text tokens); 0 = unsupported: sizes the
// MdmTextCache.sd_kfold / sd_cb / sd_vfold buffers ([L2][B][passes][128][D], [L2][B][passes][128], [L2][B][passes][D][128])
int mdm_sd_fold_passes(int32_t D, int32_t H, int32_t N) { return sd_fold_policy(D, H, N); }

int mdm_route_dump(int32_t* buf, int64_t capacity) {
  if (buf && capacity <= 0) return MDM_ERR_ARG;
  g_route_dump = buf, g_route_cap = buf ? capacity : 0;
  return MDM_OK;
}

int mdm_probe_enable(int32_t enable) {
  if (enable && !g_probe.made) {
    for (int i = 0; i < PROBE_MAX; ++i)
      if (hipEventCreate(&g_probe.a[i]) != hipSuccess || hipEventCreate(&g_probe.b[i]) != hipSuccess) return MDM_ERR_LAUNCH;
    g_probe.made = true;
  }
  g_probe.on = enable != 0;
  if (enable) g_probe.n = 0;
  return MDM_OK;
}

int mdm_probe_read(float* us, int32_t* rows, int32_t cap) {
  const int n = g_probe.n;
  for (int i = 0; i < n && i < cap; ++i) {
    if (hipEventSynchronize(g_probe.b[i]) != hipSuccess) return -1;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_probe.a[i], g_probe.b[i]) != hipSuccess) return -1;
    if (us) us[i] = ms * 1e3f;
    if (rows) rows[i] = g_probe.rows[i];
  }
  return n;
}

int mdm_cfg_posterior_step(const float* x, const float* eps_c, const float* eps_u, const float* noise, int64_t n,
                           const float* tab, int32_t steps, const int32_t* t_dev, int32_t t_imm, float cfg_scale,
                           int32_t clip_denoised, float* x_out, float* x0_out, void* stream) {
  if (steps <= 0 || (!t_dev && (t_imm < 0 || t_imm >= steps))) return MDM_ERR_ARG;
  return cfg_step(x, eps_c, eps_u, noise, n, tab, steps, t_dev, t_imm, cfg_scale, clip_denoised, x_out, x0_out,
                  (hipStream_t)stream);
}

int mdm_ddim_step(const float* x, const float* eps, const float* noise, int64_t n, const float* tab, int32_t steps,
                  const int32_t* t_dev, int32_t t_imm, float eta, int32_t clip_denoised, float* x_out, float* x0_out,
                  void* stream) {
  if (steps <= 0 || (!t_dev && (t_imm < 0 || t_imm >= steps))) return MDM_ERR_ARG;
  return ddim_step(x, eps, noise, n, tab, steps, t_dev, t_imm, eta, clip_denoised, x_out, x0_out, (hipStream_t)stream);
}

int mdm_xattn_gate(const float* gate, const float* adaptive_gate, int32_t D, float* out, void* stream) {
  if (!gate || !adaptive_gate || !out || D <= 0) return MDM_ERR_ARG;
  return xattn_gate(gate, adaptive_gate, D, out, (hipStream_t)stream);
}

}  // extern "C"
