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

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kL = 4;
constexpr int kK[kL] = {128, 64, 32, 16};
constexpr int kN[kL] = {64, 32, 16, 8};
constexpr int kP = 64;        // head: extra (GMF) columns
constexpr int kNL = 8;        // head: last activations
constexpr int kHeadW = kP + kNL;

constexpr int blocks(int units) { return (units + 15) / 16; }
// float offset of layer l's operand-ordered weights: [out block][in block][lane][chunk]
constexpr int wa_off(int l) {
  int o = 0;
  for (int i = 0; i < l; ++i) o += blocks(kN[i]) * blocks(kK[i]) * 256;
  return o;
}
constexpr int kWFloats = wa_off(kL);   // 11008
constexpr int b_off(int l) {
  int o = 0;
  for (int i = 0; i < l; ++i) o += blocks(kN[i]) * 16;
  return o;
}
constexpr int kBFloats = b_off(kL);    // 128 (biases padded to whole blocks)

struct Tower {
  const float* w[kL];
  const float* b[kL];
  float* y[kL];
  int64_t ldy[kL];
};
struct HeadFwd {
  const float* x; int64_t ldx;   // (m, >= 64) extra columns
  const float* w;                // 64 + 8 weights
  const float* c;                // one bias
  float* out; int64_t ldout; int act;
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void stg4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }

// forward A operands of every layer into LDS: s_w[wa_off(l) + ((b*J + j)*64 + lane)*4 + c] = W_l[16b + lane%16][16j + 4(lane/16) + c]
// (rows past n: zero -- their outputs stay relu(0) = 0 and are never stored).  A dwordx4 of a weight row (inputs
// 4t .. 4t+3) is one dwordx4 of the operand order (same block j, same lane, chunks 0..3): coalesced loads, all of a
// thread's ~11 in flight before the first LDS store (one element at a time this was 43 dependent round trips).
__device__ __forceinline__ void stage_forward_weights(float* s_w, float* s_b, const Tower& T) {
  constexpr int kUnits = kWFloats / 4;                       // dwordx4 units over all layers
  constexpr int kPer = (kUnits + kThreads - 1) / kThreads;   // 11
  f32x4 v[kPer];
  int dst[kPer];
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    int u = threadIdx.x + i * kThreads;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    dst[i] = -1;
    if (u < kUnits) {
      int l = 0;
#pragma unroll
      for (int t = 0; t < kL - 1; ++t)
        if (u >= wa_off(t + 1) / 4) l = t + 1;
      u -= wa_off(l) / 4;
      const int K4 = kK[l] / 4, J = blocks(kK[l]);
      const int unit = u / K4, t4 = u - unit * K4;            // row `unit` (padded to whole blocks), inputs 4 t4 ..
      const int b = unit >> 4, j = t4 >> 2, qq = t4 & 3;
      dst[i] = wa_off(l) + ((b * J + j) * 64 + qq * 16 + (unit & 15)) * 4;
      if (unit < kN[l]) v[i] = ldg4(T.w[l] + (int64_t)unit * kK[l] + 4 * t4);
    }
  }
#pragma unroll
  for (int i = 0; i < kPer; ++i)
    if (dst[i] >= 0) *reinterpret_cast<f32x4*>(s_w + dst[i]) = v[i];
#pragma unroll
  for (int l = 0; l < kL; ++l)
    for (int i = threadIdx.x; i < blocks(kN[l]) * 16; i += blockDim.x)
      s_b[b_off(l) + i] = (i < kN[l] && T.b[l]) ? T.b[l][i] : 0.0f;
}

template <int L, int NIN>   // one layer: NIN input blocks in registers -> blocks(kN[L]) output blocks
__device__ __forceinline__ void layer_fwd(const float* s_w, const float* s_b, int lane, int q, const f32x4 (&in)[NIN],
                                          f32x4 (&out)[blocks(kN[L])]) {
  constexpr int B = blocks(kN[L]), J = blocks(kK[L]);
  static_assert(J == NIN, "input blocks");
#pragma unroll
  for (int b = 0; b < B; ++b) out[b] = *reinterpret_cast<const f32x4*>(s_b + b_off(L) + 16 * b + 4 * q);
  // (input block j, output block b) steps, B independent accumulator chains interleaved; the operand of step s+2 is
  // requested while step s multiplies.  The compiler otherwise either hoists every weight read of the layer (344
  // registers, one wave per SIMD) or, held to a register budget, reads each one right in front of its use.
  constexpr int S = J * B;
  const float* wbase = s_w + wa_off(L) + lane * 4;
  auto wread = [&](int s2) { return *reinterpret_cast<const f32x4*>(wbase + (((s2 % B) * J + s2 / B) * 64) * 4); };
  f32x4 w0 = wread(0), w1 = S > 1 ? wread(1) : w0;
#pragma unroll
  for (int s2 = 0; s2 < S; ++s2) {
    const int j = s2 / B, b = s2 % B;
    const f32x4 wa = w0;
    w0 = w1;
    if (s2 + 2 < S) w1 = wread(s2 + 2);
    asm volatile("" ::: "memory");   // later weight reads stay behind this point
#pragma unroll
    for (int c = 0; c < 4; ++c) out[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c], in[j][c], out[b], 0, 0, 0);
  }
#pragma unroll
  for (int b = 0; b < B; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[b][r] = fmaxf(out[b][r], 0.0f);
}

__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3)))   // <= 168 registers: 3 workgroups per CU
ncf16_fwd_kernel(const Tower T, const float* __restrict__ x, int64_t ldx, int64_t m, const HeadFwd H) {
  __shared__ __attribute__((aligned(16))) float s_w[kWFloats];
  __shared__ __attribute__((aligned(16))) float s_b[kBFloats];
  __shared__ __attribute__((aligned(16))) float s_hw[kHeadW];
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int64_t groups = (m + 15) / 16;
  const int64_t wave0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * kThreads) >> 6;
  // the first group's rows are requested before the weights are staged
  f32x4 xb[8], xn[8];
  auto fetch = [&](int64_t g, f32x4 (&dst)[8]) {
    const int64_t row = g * 16 + n;
    const bool live = g < groups && row < m;
    const float* src = x + (live ? row : 0) * ldx + 4 * q;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = live ? ldg4(src + 16 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  fetch(wave0, xb);
  stage_forward_weights(s_w, s_b, T);
  for (int i = threadIdx.x; i < kHeadW; i += blockDim.x) s_hw[i] = H.w[i];
  __syncthreads();
  const float hc = H.c ? H.c[0] : 0.0f;
  for (int64_t g = wave0; g < groups; g += nwaves) {
    const int64_t row = g * 16 + n;
    const bool live = row < m;
    // the head's extra columns of this sample: columns 16q .. 16q+15
    f32x4 xe[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xe[i] = live ? ldg4(H.x + row * H.ldx + 16 * q + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
    fetch(g + nwaves, xn);
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
  }
}

}  // namespace

// internal entry (mlp_fused.hip dispatches here for the pinned tower + 64-column head); every pointer checked there
int ctr_ncf16_fwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, const ctr_mlp_head_t* head,
                  hipStream_t st) {
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
  if (grid > 256 * 3) grid = 256 * 3;   // three resident workgroups per CU (45 KB of LDS, <= 168 registers)
  hipLaunchKernelGGL(ncf16_fwd_kernel, dim3((unsigned)grid), dim3(kThreads), 0, st, T, x, ldx, m, H);
  return ctr_launch_status();
}
