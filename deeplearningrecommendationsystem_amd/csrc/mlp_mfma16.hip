// NeuralCF's tower + folded head (model/neuralcf.py:46-56 at BASELINE configs[1]: 128 -> 64 -> 32 -> 16 -> 8, ReLU, then
// the single-unit head on [gmf | h]) with the activations kept in MATRIX-CORE OPERAND LAYOUT from layer to layer.
//
// mlp_fused.hip walks a 32-row tile through the layers with the activations in wave-private LDS tiles: every layer
// parks its input, reads fragments back, writes its output tile, and the cycle stamps show that this fixed per-layer
// work -- not the matrix instructions -- is most of a tile's time (a 16 -> 8 layer costs as much as 64 -> 32).
// Here a wave owns SIXTEEN samples and computes Y^T (units x samples) = W (units x inputs) . X^T (inputs x samples)
// with v_mfma_f32_16x16x4_f32.  The result layout of that instruction -- lane (q, n) holds units 4q .. 4q+3 of sample
// n for every block of 16 units -- is also a legal B-operand layout for the next product if contraction chunk c of
// input block j is taken to be the units {16j + 4q + c}: bias and ReLU are applied to the accumulator registers and
// those registers ARE the next layer's operands.  No activation ever visits LDS, there is no wave barrier, and the
// weights enter as A operands read from an LDS copy that is permuted once per workgroup into operand order (one
// ds_read_b128 per four matrix instructions).  The same trick carries the GRU (gru.hip).
//
// Backward keeps the gradient in the same layout (dX^T = W^T . gZ^T from a transposed weight copy), and forms the
// weight gradients dW[i][k] = sum_n gZ[i][n] X[k][n] -- a contraction over the SAMPLES -- from unit-major operands:
// the saved activations are loaded a second time in that order, the gradient blocks are transposed through a small
// LDS strip private to the wave.  All dW blocks of the tower (172 accumulator registers) stay in the matrix
// accumulators across every sample group a wave walks; one partial per workgroup goes to the workspace in the slab
// layout of ctr_mlp_head_bwd and reduce.hip adds the partials up.
#include "ctr_common.h"

#include <stdlib.h>

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kL = 4;
constexpr int kK[kL] = {128, 64, 32, 16};
constexpr int kN[kL] = {64, 32, 16, 8};
constexpr int kP = 64;        // head: extra (GMF) columns
constexpr int kNL = 8;        // head: last activations
constexpr int kHeadW = kP + kNL;


#include "mfma16_tower.inc"

struct HeadFwd {
  const float* x; int64_t ldx;   // (m, >= 64) extra columns
  const float* w;                // 64 + 8 weights
  const float* c;                // one bias
  float* out; int64_t ldout; int act;
};

// The embedding stage in front of the tower (NeuralCF: x = [MLP_U[u] | MLP_I[i]], extra columns = GMF_U[u] * GMF_I[i],
// reference model/neuralcf.py:37-47).  GATHER: the kernel reads the ids and the table rows itself (the tables of the
// BASELINE shape sit in L2) and writes x and the product into the activation matrix on the way -- the backward and the
// embedding backward read them there -- instead of reading what a gather kernel wrote a moment earlier.
struct Gather {
  const int64_t* uidx; int64_t ustride;
  const int64_t* iidx; int64_t istride;
  const float* mlp_u; const float* mlp_i;    // (vocab, 64) each
  const float* gmf_u; const float* gmf_i;    // (vocab, 64) each
  int64_t nu, ni;
  float* xout;                               // (m, ldx): columns 0..127 of the activation matrix
  float* xeout; int64_t ldxe;                // (m, ldxe): the 64 extra columns
  int32_t* err_flag;                         // nullable: set on an id outside its table (the row read is row 0)
  int write_x;                               // forward: 0 = the backward gathers x again (ctr_embed_mlp_head_bwd), x is not written
  // forward, optional (fold_u != NULL): the head's weights are formed here instead of read from H.w / H.c --
  // ctr_fold_head_fwd's map for p = 64, n = 64, k = 8: wfold = [u[:64] | W^T u[64:]], cfold = b . u[64:] + b2 --
  // and workgroup 0 leaves them in wfold_out / cfold_out for the backward
  const float* fold_u; const float* fold_w; int64_t fold_ldw; const float* fold_b; const float* fold_b2;
  float* wfold_out; float* cfold_out;
  // backward, optional: a buffer to clear before anything of this call accumulates into it (the step's gradient
  // buffers: the zero fill was a launch of its own)
  float* zero_buf; int64_t zero_floats;      // 16-byte aligned, a multiple of 4 floats
};

template <bool GATHER>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3)))   // <= 168 registers: 3 workgroups per CU
ncf16_fwd_kernel(const Tower T, const float* __restrict__ x, int64_t ldx, int64_t m, const HeadFwd H, const Gather G) {
  __shared__ __attribute__((aligned(16))) float s_w[kWFloats];
  __shared__ __attribute__((aligned(16))) float s_b[kBFloats];
  __shared__ __attribute__((aligned(16))) float s_hw[kHeadW];
  __shared__ float s_hc;
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int64_t groups = (m + 15) / 16;
  const int64_t wave0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * kThreads) >> 6;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // the first group's rows are requested before the weights are staged
  f32x4 xb[8], xn[8];
  f32x4 xeb[4], xen[4];                      // GATHER: the product columns 16q .. 16q+15 of this lane's sample
  int64_t un = 0, in_ = 0;                   // GATHER: ids of the group after the one being fetched
  auto fetch_ids = [&](int64_t g) {
    if constexpr (GATHER) {
      const int64_t row = g * 16 + n;
      un = in_ = 0;
      if (g < groups && row < m) {
        un = G.uidx[row * G.ustride];
        in_ = G.iidx[row * G.istride];
      }
    }
  };
  auto fetch = [&](int64_t g, f32x4 (&dst)[8], f32x4 (&dxe)[4]) {
    const int64_t row = g * 16 + n;
    const bool live = g < groups && row < m;
    if constexpr (GATHER) {
      int64_t u = un, i = in_;               // requested a group earlier
      if (u < 0 || u >= G.nu) { u = 0; if (live && G.err_flag) *G.err_flag = 1; }
      if (i < 0 || i >= G.ni) { i = 0; if (live && G.err_flag) *G.err_flag = 1; }
      const float* ru = G.mlp_u + u * 64 + 4 * q;
      const float* ri = G.mlp_i + i * 64 + 4 * q;
      const float* pu = G.gmf_u + u * 64 + 16 * q;
      const float* pi = G.gmf_i + i * 64 + 16 * q;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dst[j] = live ? ldg4(ru + 16 * j) : zero4;
        dst[4 + j] = live ? ldg4(ri + 16 * j) : zero4;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) dxe[j] = live ? ldg4(pu + 4 * j) * ldg4(pi + 4 * j) : zero4;
    } else {
      const float* src = x + (live ? row : 0) * ldx + 4 * q;
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = live ? ldg4(src + 16 * j) : zero4;
    }
  };
  // ids first, the weight requests behind them, then the first group's rows (which wait for the ids alone: loads
  // return in order), the next group's ids, and only then the LDS stores that wait for the weights
  fetch_ids(wave0);
  {
    f32x4 wv[kFStagePer];
    int wdst[kFStagePer];
    float bv;
    stage_forward_load(T, wv, wdst, bv);
    float hw = 0.0f;
    if (GATHER && G.fold_u) {
      const int t = threadIdx.x;
      if (t < kP) {
        hw = G.fold_u[t];
      } else if (t < kHeadW) {                 // column t - 64 of W^T u: 64 terms, eight loads in flight
        const float* wc = G.fold_w + (t - kP);
        const float* u = G.fold_u + kP;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i0 = 0; i0 < 64; i0 += 8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] = fmaf(wc[(int64_t)(i0 + e) * G.fold_ldw], u[i0 + e], acc[e]);
        }
        hw = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
      } else if (t == kHeadW) {                // the folded bias
        const float* u = G.fold_u + kP;
        float acc = G.fold_b2 ? G.fold_b2[0] : 0.0f;
        if (G.fold_b)
          for (int i = 0; i < 64; ++i) acc = fmaf(G.fold_b[i], u[i], acc);
        hw = acc;
      }
      if (blockIdx.x == 0) {
        if (t < kHeadW) G.wfold_out[t] = hw;
        if (t == kHeadW) G.cfold_out[0] = hw;
      }
    } else if (threadIdx.x < kHeadW) {
      hw = H.w[threadIdx.x];
    }
    fetch(wave0, xb, xeb);
    fetch_ids(wave0 + nwaves);
    stage_forward_store(s_w, s_b, wv, wdst, bv);
    static_assert(kHeadW <= kThreads, "one head weight per thread");
    if (threadIdx.x < kHeadW) s_hw[threadIdx.x] = hw;
    if (GATHER && G.fold_u && threadIdx.x == kHeadW) s_hc = hw;
  }
  __syncthreads();
  const float hc = (GATHER && G.fold_u) ? s_hc : (H.c ? H.c[0] : 0.0f);
  for (int64_t g = wave0; g < groups; g += nwaves) {
    const int64_t row = g * 16 + n;
    const bool live = row < m;
    // the head's extra columns of this sample: columns 16q .. 16q+15
    f32x4 xe[4];
    if constexpr (GATHER) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xe[i] = xeb[i];
      if (live) {
        if (G.write_x) {
#pragma unroll
          for (int j = 0; j < 8; ++j) stg4(G.xout + row * ldx + 16 * j + 4 * q, xb[j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) stg4(G.xeout + row * G.ldxe + 16 * q + 4 * i, xe[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) xe[i] = live ? ldg4(H.x + row * H.ldx + 16 * q + 4 * i) : zero4;
    }
    fetch(g + nwaves, xn, xen);
    fetch_ids(g + 2 * nwaves);
    f32x4 y1[4], y2[2], y3[1], y4[1];
    layer_fwd<0, 8>(s_w, s_b, lane, q, xb, y1);
    if (live) {
#pragma unroll
      for (int b = 0; b < 4; ++b) stg4(T.y[0] + row * T.ldy[0] + 16 * b + 4 * q, y1[b]);
    }
    layer_fwd<1, 4>(s_w, s_b, lane, q, y1, y2);
    if (live) {
#pragma unroll
      for (int b = 0; b < 2; ++b) stg4(T.y[1] + row * T.ldy[1] + 16 * b + 4 * q, y2[b]);
    }
    layer_fwd<2, 2>(s_w, s_b, lane, q, y2, y3);
    if (live) stg4(T.y[2] + row * T.ldy[2] + 4 * q, y3[0]);
    layer_fwd<3, 1>(s_w, s_b, lane, q, y3, y4);
    if (live && q < 2) stg4(T.y[3] + row * T.ldy[3] + 4 * q, y4[0]);
    // head: prob = act([gmf | h] . w + c); the four lanes of a sample hold 16 + (q < 2 ? 4 : 0) terms each
    float dot = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(s_hw + 16 * q + 4 * i);
      dot = fmaf(xe[i][0], wv[0], dot); dot = fmaf(xe[i][1], wv[1], dot);
      dot = fmaf(xe[i][2], wv[2], dot); dot = fmaf(xe[i][3], wv[3], dot);
    }
    if (q < 2) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(s_hw + kP + 4 * q);
      dot = fmaf(y4[0][0], wv[0], dot); dot = fmaf(y4[0][1], wv[1], dot);
      dot = fmaf(y4[0][2], wv[2], dot); dot = fmaf(y4[0][3], wv[3], dot);
    }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    if (q == 0 && live) H.out[row * H.ldout] = ctr_act(dot + hc, H.act);
#pragma unroll
    for (int j = 0; j < 8; ++j) xb[j] = xn[j];
    if constexpr (GATHER) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xeb[i] = xen[i];
    }
  }
}

// ------------------------------------------------------------------ backward
struct HeadBwd {
  const float* gprob; int64_t ldgp;
  const float* prob; int64_t ldp;
  const float* xe; int64_t ldxe;     // the 64 extra columns
  const float* w;                    // 64 + 8 head weights
  float* gxe; int64_t ldgxe;         // gradient of the extra columns (=)
  int act;
};

constexpr int kStrip = 6 * kTile;                         // per wave: tiles A0 A1 | B0..B3 (layers alternate)
constexpr int kBwdMain = kWFloats + kWaves * kStrip;      // floats: transposed weights, then the waves' strips
constexpr int slab_w(int l) {                             // float offset of layer l's dW inside a slab
  int o = 0;
  for (int i = 0; i < l; ++i) o += kN[i] * kK[i] + kN[i];
  return o;
}
constexpr int kSlabHead = slab_w(kL);                     // 11000
constexpr int kSlab = kSlabHead + kHeadW + 1;             // 11073
// at the end three waves park their sums side by side (the weights are dead), the fourth adds into the first copy;
// a copy = the 43 accumulator vectors in lane order, then the bias sums of the four layers and the head's sums
constexpr int kVecs = 43;
constexpr int kVecFloats = kVecs * 256;
constexpr int small_b(int l) {                            // bias sums of layer l (l = 4: the head's sums) inside "small"
  int o = 0;
  for (int i = 0; i < l; ++i) o += kN[i];
  return o;
}
constexpr int kSmall = small_b(kL) + kHeadW + 1;          // 193
constexpr int kSmallVecs = 13;                            // bias sums 4 + 2 + 1 + 1, head: 1 + 4 (each 4 per lane)
constexpr int kCopy = kVecFloats + kSmall + 3;            // 11204 (a multiple of 4: every copy stays 16-byte aligned)
constexpr int kBwdLds = 3 * kCopy > kBwdMain ? 3 * kCopy : kBwdMain;

#ifdef CTR_STAMPS
__device__ unsigned long long g_stamps[4 * 16];
#define STAMP(i) do { if (blockIdx.x == 7 && lane == 0) g_stamps[wave * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// GATHER: the tower's input is not read from memory but gathered again from the two MLP tables by the samples' ids
// (the forward then never wrote it: 33 MB less to write there, 33 MB less to read here; the tables sit in L2).
template <bool GATHER>
__global__ void __launch_bounds__(kThreads)
ncf16_bwd_kernel(const Tower T, const float* __restrict__ x, int64_t ldx, int64_t m, const HeadBwd H,
                 float* __restrict__ gx, int64_t ldgx, float* __restrict__ ws, const Gather G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ __attribute__((aligned(16))) float s_hw[kHeadW];
  float* s_wt = lds;
  const int lane = threadIdx.x & 63, q = lane >> 4, lo = lane & 15, wave = threadIdx.x >> 6;
  float* tA = lds + kWFloats + wave * kStrip;   // tiles of layers 3 and 1
  float* tB = tA + 2 * kTile;                    // tiles of layers 2 and 0
  const int64_t groups = (m + 15) / 16;
  const int64_t wave0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * kThreads) >> 6;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- operands of a sample group, all requested together (a group ahead of their use)
  float gp = 0.0f, pb = 0.0f;
  f32x4 y4d = zero4, xe[4], y3d = zero4, y2d[2], y1d[4];     // sample-major: units 4q + r (+16 per block) of sample lo
  f32x4 y3t = zero4, y2t[2], y1t[4], x0t[8];                 // unit-major: unit lo (+16 per block) of samples 4q + c
  // Requested per layer, for the NEXT group, as soon as this group has consumed the registers: every load then has
  // most of a group's time (~10 us) to land.  (All 72 requested together in front of dX_0 left 37 % of the wave's life
  // in s_waitcnt: every wave of the chip issued its burst at the same moment.)
  // Row offsets are 32-bit (the entry point checks m * ld < 2^30 for every operand): one multiply per address in
  // place of a 64-bit product, and the loads take the "scalar base + 32-bit lane offset" form.
  // the rows of the group being requested: set once per group by rows_for, used by all five fetch_* of that group
  uint32_t rc = 0, rt[4] = {0, 0, 0, 0};
  uint32_t uo[4] = {0, 0, 0, 0}, io[4] = {0, 0, 0, 0};
  bool rlive = false;
  auto rows_for = [&](int64_t g) {
#ifdef CTR_STAMPS_HOT
    g &= 63;   // timing experiment: every wave reads cache-resident rows (results are wrong)
#endif
    const int64_t row = g * 16 + lo;
    rlive = g < groups && row < m;
    rc = (uint32_t)(rlive ? row : (m - 1));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t r = g * 16 + 4 * q + c;
      rt[c] = (uint32_t)((g >= groups || r >= m) ? (m - 1) : r);   // (a clamped row meets a zero gradient)
    }
    if constexpr (GATHER) {
      // byte offsets of the four samples' rows in the two tables (ids outside a table read row 0, as in the forward);
      // requested at the head of a group, used by fetch_l0 at its end
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int64_t u = G.uidx[(int64_t)rt[c] * G.ustride], i = G.iidx[(int64_t)rt[c] * G.istride];
        if (u < 0 || u >= G.nu) u = 0;
        if (i < 0 || i >= G.ni) i = 0;
        uo[c] = (uint32_t)u * 256u;
        io[c] = (uint32_t)i * 256u;
      }
    }
  };
  auto at = [](const float* base, uint32_t row, uint32_t ld, uint32_t col) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (row * ld + col) * 4u);
  };
  const uint32_t ld0 = (uint32_t)T.ldy[0], ld1 = (uint32_t)T.ldy[1], ld2 = (uint32_t)T.ldy[2], ld3 = (uint32_t)T.ldy[3];
  const uint32_t ldx32 = (uint32_t)ldx, ldxe = (uint32_t)H.ldxe;
  auto fetch_head = [&]() {
    gp = rlive ? *at(H.gprob, rc, (uint32_t)H.ldgp, 0) : 0.0f;   // a dead lane's gz is zero: it adds nothing anywhere
    pb = *at(H.prob, rc, (uint32_t)H.ldp, 0);
    y4d = q < 2 ? ldg4(at(T.y[3], rc, ld3, 4 * q)) : zero4;
#pragma unroll
    for (int i = 0; i < 4; ++i) xe[i] = ldg4(at(H.xe, rc, ldxe, 16 * q + 4 * i));
  };
  auto fetch_l3 = [&]() {
    y3d = ldg4(at(T.y[2], rc, ld2, 4 * q));
#pragma unroll
    for (int c = 0; c < 4; ++c) y3t[c] = *at(T.y[2], rt[c], ld2, lo);
  };
  auto fetch_l2 = [&]() {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      y2d[b] = ldg4(at(T.y[1], rc, ld1, 16 * b + 4 * q));
#pragma unroll
      for (int c = 0; c < 4; ++c) y2t[b][c] = *at(T.y[1], rt[c], ld1, 16 * b + lo);
    }
  };
  auto fetch_l1 = [&]() {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      y1d[b] = ldg4(at(T.y[0], rc, ld0, 16 * b + 4 * q));
#pragma unroll
      for (int c = 0; c < 4; ++c) y1t[b][c] = *at(T.y[0], rt[c], ld0, 16 * b + lo);
    }
  };
  auto fetch_l0 = [&]() {
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (GATHER)
          x0t[b][c] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(b < 4 ? G.mlp_u : G.mlp_i) +
                                                     ((b < 4 ? uo[c] : io[c]) + (16 * (b & 3) + lo) * 4u));
        else
          x0t[b][c] = *at(x, rt[c], ldx32, 16 * b + lo);
      }
  };
  STAMP(0);
  if (GATHER && G.zero_buf) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 4; i < G.zero_floats; i += (int64_t)gridDim.x * kThreads * 4)
      stg4(G.zero_buf + i, z);
  }
  // ---- what a lane sums over every group it walks (zeroed first: 350 moves that then pass under the first loads)
  f32x4 dw0[4][8], dw1[2][4], dw2[2], dw3;                   // dW blocks: register r = row 4q + r, column lo
  f32x4 sb0[4], sb1[2], sb2, sb3;                            // bias sums of this lane's sample: units 4q + r
  f32x4 hy = zero4, hx[4];
  float hc = 0.0f;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    sb0[b] = zero4;
    hx[b] = zero4;
#pragma unroll
    for (int j = 0; j < 8; ++j) dw0[b][j] = zero4;
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    sb1[b] = zero4;
    dw2[b] = zero4;
#pragma unroll
    for (int j = 0; j < 4; ++j) dw1[b][j] = zero4;
  }
  sb2 = sb3 = dw3 = zero4;
  {
    f32x4 wv[kStagePer];
    int wdst[kStagePer];
    stage_transposed_load(T, wv, wdst);
    STAMP(12);
    // (a wave has at most 63 vector-memory operations in flight: with all 84 requested here the issue itself stalled
    // on the first cold loads for 6 K cycles.  x of the first group is needed last: requested behind the stores.)
    rows_for(wave0);
    fetch_head(); fetch_l3(); fetch_l2(); fetch_l1();
    STAMP(13);
    stage_transposed_store(s_wt, wv, wdst);
    fetch_l0();
  }
  STAMP(14);
  if (threadIdx.x < kHeadW) s_hw[threadIdx.x] = H.w[threadIdx.x];
  __syncthreads();
  STAMP(1);

  STAMP(2);
  // ---- the walk.  Per layer: dX products first (their weights stream from LDS, the first pair requested ahead), the
  // next layer's gradient goes to the other tile set, then the dW products cover that LDS round trip.  The operands of
  // the next group are requested as registers free.  Consecutive matrix-core instructions never share an accumulator
  // where the shape allows (a dependent v_mfma_f32_16x16x4_f32 issues after 40 cycles, an independent one after 32).
  // (Threading the next group's head and layers 3 / 2 through the dW_0 products was measured too: the vector and LDS
  // instructions cost the same cycles between matrix-core instructions as in front of them, and the extra live state
  // spilled: DESIGN.md section 9.)
  int it = 0;
  for (int64_t g = wave0; g < groups; g += nwaves, ++it) {
    if (it < 4) STAMP(3 + it);
    const int64_t row = g * 16 + lo;
    const bool live = row < m;
    // ---- head: gz, the extra columns' gradient, the head's sums, the tower's (masked) gY
    const float gz = gp * ctr_act_grad(pb, H.act);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(s_hw + 16 * q + 4 * i);
      if (live) stg4(H.gxe + row * H.ldgxe + 16 * q + 4 * i, gz * wv);
      hx[i] += gz * xe[i];
    }
    if (q == 0) hc += gz;
    f32x4 gz3[1];
    {
      const f32x4 wv = q < 2 ? *reinterpret_cast<const f32x4*>(s_hw + kP + 4 * q) : zero4;
      hy += gz * y4d;                                         // (y4d is zero for q >= 2)
      gz3[0] = relu_mask(gz * wv, y4d);
    }
    sb3 += gz3[0];
    tiles_put<1>(tA, q, lo, gz3);
    f32x4 w0 = zero4, w1 = zero4;
    dx_first<3>(s_wt, lane, w0, w1);
    rows_for(g + nwaves);   // every request from here to the end of this group is for the next one
    fetch_head();
    // ---- layer 3 (16 -> 8)
    f32x4 gz2[1];
    {
      f32x4 tg[1];
      tiles_get<1>(tA, q, lo, tg);
      dx_layer<3, 1>(s_wt, lane, gz3, w0, w1, [&](int, const f32x4& d0, const f32x4&) { gz2[0] = relu_mask(d0, y3d); });
      sb2 += gz2[0];
      tiles_put<1>(tB, q, lo, gz2);
      dx_first<2>(s_wt, lane, w0, w1);
#pragma unroll
      for (int c = 0; c < 4; ++c) dw3 = __builtin_amdgcn_mfma_f32_16x16x4f32(tg[0][c], y3t[c], dw3, 0, 0, 0);
    }
    fetch_l3();
    // ---- layer 2 (32 -> 16)
    f32x4 gz1[2];
    {
      f32x4 tg[1];
      tiles_get<1>(tB, q, lo, tg);
      dx_layer<2, 1>(s_wt, lane, gz2, w0, w1, [&](int, const f32x4& d0, const f32x4& d1) {
        gz1[0] = relu_mask(d0, y2d[0]);
        gz1[1] = relu_mask(d1, y2d[1]);
      });
#pragma unroll
      for (int b = 0; b < 2; ++b) sb1[b] += gz1[b];
      tiles_put<2>(tA, q, lo, gz1);
      dx_first<1>(s_wt, lane, w0, w1);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) dw2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(tg[0][c], y2t[j][c], dw2[j], 0, 0, 0);
    }
    fetch_l2();
    // ---- layer 1 (64 -> 32)
    f32x4 gz0[4];
    {
      f32x4 tg[2];
      tiles_get<2>(tA, q, lo, tg);
      dx_layer<1, 2>(s_wt, lane, gz1, w0, w1, [&](int j, const f32x4& d0, const f32x4& d1) {
        gz0[j] = relu_mask(d0, y1d[j]);
        gz0[j + 1] = relu_mask(d1, y1d[j + 1]);
        sb0[j] += gz0[j];
        sb0[j + 1] += gz0[j + 1];
      });
      tiles_put<4>(tB, q, lo, gz0);
      dx_first<0>(s_wt, lane, w0, w1);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            dw1[b][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(tg[b][c], y1t[j][c], dw1[b][j], 0, 0, 0);
    }
    fetch_l1();
    // ---- layer 0 (128 -> 64)
    {
      f32x4 tg[4];
      tiles_get<4>(tB, q, lo, tg);
      dx_layer<0, 4>(s_wt, lane, gz0, w0, w1, [&](int j, const f32x4& d0, const f32x4& d1) {
        if (live) {
          stg4(gx + row * ldgx + 16 * j + 4 * q, d0);
          stg4(gx + row * ldgx + 16 * (j + 1) + 4 * q, d1);
        }
      });
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            dw0[b][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(tg[b][c], x0t[j][c], dw0[b][j], 0, 0, 0);
    }
    fetch_l0();   // x of the next group: needed last, a whole group from now
  }

  STAMP(8);
  // ---- the workgroup's partial (the weights are dead).  A copy holds the 43 accumulator vectors exactly as the lanes
  // hold them (16 B per lane: full-rate, conflict-free LDS traffic), then the 193 bias / head sums; waves 0..2 write three
  // copies at once, wave 3 adds into the first, and the copies are summed and put into slab order on the way out.
  // (Stored in slab order the rows of a quarter-wave were 512 floats apart: four-way bank conflicts on 690 scalar
  // stores per lane, and wave 3's read-modify-write of them, were 30 % of the kernel.)
  __syncthreads();
  {
    // sums over the samples (the sixteen lanes of a DPP row), four independent chains at a time
    auto rsum = [&](const f32x4& v) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = row_sum16(v[r]);
      return o;
    };
    f32x4 sm[kSmallVecs];
#pragma unroll
    for (int b = 0; b < 4; ++b) sm[b] = rsum(sb0[b]);
#pragma unroll
    for (int b = 0; b < 2; ++b) sm[4 + b] = rsum(sb1[b]);
    sm[6] = rsum(sb2);
    sm[7] = rsum(sb3);
    sm[8] = rsum(hy);
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[9 + i] = rsum(hx[i]);
    const float vc = row_sum16(hc);
    // where lane (q, 0) keeps vector i of the small part (16-byte aligned; -1: not this quarter)
    auto small_at = [&](int i) {
      return i < 4 ? small_b(0) + 16 * i + 4 * q : i < 6 ? small_b(1) + 16 * (i - 4) + 4 * q
           : i == 6 ? small_b(2) + 4 * q : i == 7 ? (q < 2 ? small_b(3) + 4 * q : -1)
           : i == 8 ? (q < 2 ? small_b(4) + kP + 4 * q : -1) : small_b(4) + 16 * q + 4 * (i - 9);
    };
    float* copy = lds + (wave < 3 ? wave : 0) * kCopy;
    float* small = copy + kVecFloats;
    auto vec = [&](int v) -> const f32x4& {
      return v < 32 ? dw0[v >> 3][v & 7] : v < 40 ? dw1[(v - 32) >> 2][(v - 32) & 3] : v < 42 ? dw2[v - 40] : dw3;
    };
    if (wave < 3) {
#pragma unroll
      for (int v = 0; v < kVecs; ++v) *reinterpret_cast<f32x4*>(copy + (v * 64 + lane) * 4) = vec(v);
      if (lo == 0) {
#pragma unroll
        for (int i = 0; i < kSmallVecs; ++i)
          if (small_at(i) >= 0) *reinterpret_cast<f32x4*>(small + small_at(i)) = sm[i];
      }
      if (lane == 0) small[small_b(4) + kHeadW] = vc;
    }
    __syncthreads();
    STAMP(9);
    if (wave == 3) {
      // read-modify-write of the first copy, the reads a dozen at a time (one at a time each waited out its own
      // LDS round trip: 12 K cycles for 56 of them)
      constexpr int kBatch = 11;
#pragma unroll
      for (int v0 = 0; v0 < kVecs; v0 += kBatch) {
        f32x4 old[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i)
          if (v0 + i < kVecs) old[i] = *reinterpret_cast<const f32x4*>(copy + ((v0 + i) * 64 + lane) * 4);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < kBatch; ++i)
          if (v0 + i < kVecs) *reinterpret_cast<f32x4*>(copy + ((v0 + i) * 64 + lane) * 4) = old[i] + vec(v0 + i);
        asm volatile("" ::: "memory");
      }
      if (lo == 0) {
        f32x4 old[kSmallVecs];
#pragma unroll
        for (int i = 0; i < kSmallVecs; ++i)
          if (small_at(i) >= 0) old[i] = *reinterpret_cast<const f32x4*>(small + small_at(i));
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < kSmallVecs; ++i)
          if (small_at(i) >= 0) *reinterpret_cast<f32x4*>(small + small_at(i)) = old[i] + sm[i];
      }
      if (lane == 0) small[small_b(4) + kHeadW] += vc;
    }
    __syncthreads();
    STAMP(10);
  }
  float* out = ws + (int64_t)blockIdx.x * kSlab;
#pragma unroll
  for (int i = 0; i < (kVecs + kWaves - 1) / kWaves; ++i) {
    const int v = wave + kWaves * i;                          // uniform over the wave
    if (v < kVecs) {
      const int at = (v * 64 + lane) * 4;
      const f32x4 sum = (*reinterpret_cast<const f32x4*>(lds + at) + *reinterpret_cast<const f32x4*>(lds + kCopy + at)) +
                        *reinterpret_cast<const f32x4*>(lds + 2 * kCopy + at);
      const int l = v < 32 ? 0 : v < 40 ? 1 : v < 42 ? 2 : 3;
      const int vv = v - (l == 0 ? 0 : l == 1 ? 32 : l == 2 ? 40 : 42);
      const int J = l == 0 ? 8 : l == 1 ? 4 : l == 2 ? 2 : 1, K = l == 0 ? kK[0] : l == 1 ? kK[1] : l == 2 ? kK[2] : kK[3];
      const int bb = vv / J, jj = vv - bb * J;
      const int base = (l == 0 ? slab_w(0) : l == 1 ? slab_w(1) : l == 2 ? slab_w(2) : slab_w(3)) +
                       (16 * bb + 4 * q) * K + 16 * jj + lo;
      if (l < 3 || q < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[base + r * K] = sum[r];
      }
    }
  }
  for (int i = threadIdx.x; i < kSmall; i += kThreads) {
    const int l = i < small_b(1) ? 0 : i < small_b(2) ? 1 : i < small_b(3) ? 2 : i < small_b(4) ? 3 : 4;
    const int dst = l == 4 ? kSlabHead + (i - small_b(4)) : slab_w(l) + kN[l] * kK[l] + (i - small_b(l));
    const int at = kVecFloats + i;
    out[dst] = (lds[at] + lds[kCopy + at]) + lds[2 * kCopy + at];
  }
  STAMP(11);
}

}  // namespace

int ctr_ncf16_slab_floats() { return kSlab; }
#ifdef CTR_STAMPS
extern "C" __attribute__((visibility("default"))) int ctr_ncf16_debug_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64);
}
#endif

static int ncf16_bwd_launch(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, const ctr_mlp_head_grad_t* hg,
                            float* gx, int64_t ldgx, float* workspace, int64_t workspace_floats, int* grid_out,
                            const Gather* gather, hipStream_t st) {
  Tower T;
  for (int l = 0; l < kL; ++l) {
    if (layers[l].n != kN[l] || layers[l].k != kK[l] || layers[l].act != CTR_ACT_RELU) return CTR_ELIMIT;
    if (!ctr_aligned16(layers[l].y) || layers[l].ldy % 4 != 0 || !ctr_aligned16(layers[l].w)) return CTR_ELIMIT;
    T.w[l] = layers[l].w; T.b[l] = layers[l].b; T.y[l] = layers[l].y; T.ldy[l] = layers[l].ldy;
  }
  if (hg->p != kP || !gx || !ctr_aligned16(gx) || ldgx % 4 != 0 || !ctr_aligned16(hg->x) || hg->ldx % 4 != 0 ||
      !ctr_aligned16(hg->gx) || hg->ldgx % 4 != 0)
    return CTR_ELIMIT;
  // 32-bit byte offsets inside every operand the kernel reads
  int64_t widest = ldx > hg->ldx ? ldx : hg->ldx;
  for (int l = 0; l < kL; ++l) widest = layers[l].ldy > widest ? layers[l].ldy : widest;
  widest = hg->ldgprob > widest ? hg->ldgprob : widest;
  widest = hg->ldprob > widest ? hg->ldprob : widest;
  if (m * widest >= ((int64_t)1 << 30)) return CTR_ELIMIT;
  const int64_t groups = ctr_ceil_div(m, 16);
  int64_t grid = ctr_ceil_div(groups, kWaves);
  if (grid > 256) grid = 256;   // 172 accumulator registers per lane: one wave per SIMD, one workgroup per CU
  if (workspace_floats < grid * kSlab) return CTR_ELIMIT;
  const size_t lds_bytes = sizeof(float) * kBwdLds;
  const void* kern = gather ? reinterpret_cast<const void*>(ncf16_bwd_kernel<true>)
                             : reinterpret_cast<const void*>(ncf16_bwd_kernel<false>);
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return CTR_ELAUNCH;
  const HeadBwd H{hg->gprob, hg->ldgprob, hg->prob, hg->ldprob, hg->x, hg->ldx, hg->w, hg->gx, hg->ldgx, hg->act};
  if (gather)
    hipLaunchKernelGGL(ncf16_bwd_kernel<true>, dim3((unsigned)grid), dim3(kThreads), lds_bytes, st, T, x, ldx, m, H, gx, ldgx,
                       workspace, *gather);
  else
    hipLaunchKernelGGL(ncf16_bwd_kernel<false>, dim3((unsigned)grid), dim3(kThreads), lds_bytes, st, T, x, ldx, m, H, gx,
                       ldgx, workspace, Gather{});
  *grid_out = (int)grid;
  return ctr_launch_status();
}

int ctr_ncf16_bwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, const ctr_mlp_head_grad_t* hg,
                  float* gx, int64_t ldgx, float* workspace, int64_t workspace_floats, int* grid_out, hipStream_t st) {
  return ncf16_bwd_launch(x, ldx, m, layers, hg, gx, ldgx, workspace, workspace_floats, grid_out, nullptr, st);
}

// the NeuralCF field pattern of ctr_ncf16_gather_fwd (x = [MLP_U[u] | MLP_I[i]]); false: not that pattern
static bool ncf_pattern(const ctr_field_t* fields, int nfields, Gather* G) {
  if (nfields != 3) return false;
  const ctr_field_t &fu = fields[0], &fi = fields[1], &fp = fields[2];
  const bool pattern = fu.kind == CTR_FIELD_ID_I64 && fi.kind == CTR_FIELD_ID_I64 && fp.kind == CTR_FIELD_PROD_I64 &&
                       fu.width == 64 && fi.width == 64 && fp.width == 64 && fu.out_col == 0 && fi.out_col == 64 &&
                       fp.out_col == 128 && fp.idx == fu.idx && fp.idx2 == fi.idx && fp.idx_stride == fu.idx_stride &&
                       fp.idx_stride == fi.idx_stride && fp.vocab == fu.vocab && fp.vocab2 == fi.vocab &&
                       fu.vocab < (1 << 24) && fi.vocab < (1 << 24);   // 32-bit byte offsets of 256-byte rows
  if (!pattern || !fu.idx || !fi.idx) return false;
  const float* tabs[4] = {fu.table, fi.table, fp.table, fp.table2};
  for (const float* t : tabs)
    if (!t || !ctr_aligned16(t)) return false;
  *G = Gather{fu.idx, fu.idx_stride, fi.idx, fi.idx_stride, fu.table, fi.table, fp.table, fp.table2, fu.vocab, fi.vocab,
              nullptr, nullptr, 0, nullptr, 1, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  return true;
}

// ctr_ncf16_bwd with the tower's input gathered from the tables (see ncf16_bwd_kernel<true>); x / ldx only describe
// the row stride the forward used for the other columns and are not read
int ctr_ncf16_gather_bwd(const ctr_field_t* fields, int nfields, int64_t m, const ctr_mlp_layer_t* layers,
                         const ctr_mlp_head_grad_t* hg, float* gx, int64_t ldgx, float* workspace,
                         int64_t workspace_floats, int* grid_out, float* zero_buf, int64_t zero_floats, hipStream_t st) {
  Gather G;
  if (!ncf_pattern(fields, nfields, &G)) return CTR_ELIMIT;
  if (zero_buf && (!ctr_aligned16(zero_buf) || zero_floats % 4 != 0)) return CTR_ELIMIT;
  G.zero_buf = zero_buf;
  G.zero_floats = zero_buf ? zero_floats : 0;
  return ncf16_bwd_launch(hg->x, hg->ldx, m, layers, hg, gx, ldgx, workspace, workspace_floats, grid_out, &G, st);
}

// internal entry (mlp_fused.hip dispatches here for the pinned tower + 64-column head); every pointer checked there
static int ncf16_fwd_launch(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, const ctr_mlp_head_t* head,
                            const Gather* gather, hipStream_t st) {
  Tower T;
  for (int l = 0; l < kL; ++l) {
    if (layers[l].n != kN[l] || layers[l].k != kK[l] || layers[l].act != CTR_ACT_RELU) return CTR_ELIMIT;
    if (!ctr_aligned16(layers[l].y) || layers[l].ldy % 4 != 0) return CTR_ELIMIT;
    T.w[l] = layers[l].w; T.b[l] = layers[l].b; T.y[l] = layers[l].y; T.ldy[l] = layers[l].ldy;
  }
  if (head->p != kP || !ctr_aligned16(head->x) || head->ldx % 4 != 0 || !ctr_aligned16(x) || ldx % 4 != 0) return CTR_ELIMIT;
  const HeadFwd H{head->x, head->ldx, head->w, head->c, head->out, head->ldout, head->act};
  const int64_t groups = ctr_ceil_div(m, 16);
  int64_t grid = ctr_ceil_div(groups, kWaves);
  constexpr int wgs = 3;
  if (grid > 256 * wgs) grid = 256 * wgs;   // three resident workgroups per CU (45 KB of LDS, <= 168 registers)
  if (gather)
    hipLaunchKernelGGL(ncf16_fwd_kernel<true>, dim3((unsigned)grid), dim3(kThreads), 0, st, T, x, ldx, m, H, *gather);
  else
    hipLaunchKernelGGL(ncf16_fwd_kernel<false>, dim3((unsigned)grid), dim3(kThreads), 0, st, T, x, ldx, m, H, Gather{});
  return ctr_launch_status();
}

int ctr_ncf16_fwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, const ctr_mlp_head_t* head,
                  hipStream_t st) {
  return ncf16_fwd_launch(x, ldx, m, layers, head, nullptr, st);
}

// The embedding stage of NeuralCF and its tower in one launch: `fields` must be exactly
//   [ID_I64 64 -> col 0, ID_I64 64 -> col 64, PROD_I64 64 -> col 128] of one (m, ldo) matrix `out`
// whose first 128 columns are the tower's input and whose columns 128..191 are the head's extra columns.
// CTR_ELIMIT: not this pattern (nothing enqueued; the caller runs ctr_embed_fwd + ctr_mlp_head_fwd).
int ctr_ncf16_gather_fwd(const ctr_field_t* fields, int nfields, int64_t m, float* out, int64_t ldo, int32_t* err_flag,
                         int write_x, const ctr_mlp_layer_t* layers, const ctr_mlp_head_t* head, const ctr_head_fold_t* fold,
                         hipStream_t st) {
  Gather G;
  if (!ncf_pattern(fields, nfields, &G) || ldo < 192 || head->x != out + 128 || head->ldx != ldo) return CTR_ELIMIT;
  if (fold) {
    if (fold->p != kP || fold->n != 64 || fold->k != kNL || !fold->u_full || !fold->w || fold->ldw < fold->k ||
        fold->wfold != head->w || fold->cfold != head->c)
      return CTR_ELIMIT;
    G.fold_u = fold->u_full; G.fold_w = fold->w; G.fold_ldw = fold->ldw; G.fold_b = fold->b; G.fold_b2 = fold->b2;
    G.wfold_out = fold->wfold; G.cfold_out = fold->cfold;
  }
  G.xout = out;
  G.xeout = out + 128;
  G.ldxe = ldo;
  G.err_flag = err_flag;
  G.write_x = write_x;
  return ncf16_fwd_launch(out, ldo, m, layers, head, &G, st);
}
