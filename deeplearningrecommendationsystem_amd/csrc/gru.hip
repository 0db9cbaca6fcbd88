// DIEN interest evolution: single-layer batch_first GRU, h0 = 0 (model/dien.py:47,61).
// The input projection gi = X W_ih^T + b_ih for ALL steps is one MFMA GEMM (linear
// kernel); what is inherently sequential -- gh = W_hh h_{t-1} + b_hh, the gates and
// the state update, L dependent steps -- runs here, batched across samples: a group
// of G >= E lanes owns one sample (lane j = hidden unit j), W_hh sits in LDS with an
// odd row stride (row reads for the forward dots and column reads for the backward
// W_hh^T product are both conflict-free), the running state is exchanged through LDS.
// PyTorch gate order (r, z, n):  r = s(gi_r+gh_r), z = s(gi_z+gh_z),
// n = tanh(gi_n + r*gh_n), h' = (1-z)*n + z*h.
// Backward writes dgi_t and dgh_t for every step; the weight/bias/input gradients are
// then GEMMs over all (b,t) rows (dW_ih = dgi^T X, dW_hh = dgh^T H_prev, dX = dgi W_ih).
#include "ctr_common.h"
#include <stdlib.h>

namespace {

constexpr int kBlock = 256;

inline int pow2_ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

struct GruGeom {
  int64_t batch;
  int len, dim, group;  // group = lanes per sample (power of two >= dim)
};

__device__ __forceinline__ void load_whh(float* s_w, const float* __restrict__ w_hh, int dim) {
  const int stride = dim + 1;
  for (int i = threadIdx.x; i < 3 * dim * dim; i += blockDim.x) {
    const int r = i / dim, c = i - r * dim;
    s_w[r * stride + c] = w_hh[i];
  }
}

// hbuf: (batch, len+1, dim); hbuf[b,0,:] = 0, hbuf[b,t+1,:] = h_t
__global__ void __launch_bounds__(kBlock)
gru_fwd_kernel(const GruGeom g, const float* __restrict__ gi, int64_t ldgi, const float* __restrict__ w_hh,
               const float* __restrict__ b_hh, float* __restrict__ hbuf, float* __restrict__ last, int64_t ldl) {
  extern __shared__ float lds[];
  const int stride = g.dim + 1;
  float* s_w = lds;                                  // [3E][E+1]
  float* s_h = s_w + 3 * g.dim * stride;             // [2][samples][E]
  const int per_block = kBlock / g.group;
  const int sample = threadIdx.x / g.group, j = threadIdx.x % g.group;
  const bool unit = j < g.dim;
  load_whh(s_w, w_hh, g.dim);
  const float br = unit ? b_hh[j] : 0.f, bz = unit ? b_hh[g.dim + j] : 0.f, bn = unit ? b_hh[2 * g.dim + j] : 0.f;
  for (int64_t b0 = (int64_t)blockIdx.x * per_block; b0 < g.batch; b0 += (int64_t)gridDim.x * per_block) {
    const int64_t b = b0 + sample;
    const bool live = unit && b < g.batch;
    float h = 0.0f;
    if (unit) s_h[sample * g.dim + j] = 0.0f;
    if (live) hbuf[b * (g.len + 1) * g.dim + j] = 0.0f;
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < g.len; ++t) {
      float hn = 0.0f;
      if (live) {
        const float* hv = s_h + (cur * per_block + sample) * g.dim;
        float ar = br, az = bz, an = bn;
        const float* wr = s_w + j * stride;
        const float* wz = s_w + (g.dim + j) * stride;
        const float* wn = s_w + (2 * g.dim + j) * stride;
        for (int k = 0; k < g.dim; ++k) {
          const float hk = hv[k];
          ar = fmaf(wr[k], hk, ar);
          az = fmaf(wz[k], hk, az);
          an = fmaf(wn[k], hk, an);
        }
        const float* gir = gi + (b * g.len + t) * ldgi;
        const float r = ctr_sigmoid(gir[j] + ar);
        const float z = ctr_sigmoid(gir[g.dim + j] + az);
        const float n = tanhf(gir[2 * g.dim + j] + r * an);
        hn = (1.0f - z) * n + z * h;
        hbuf[(b * (g.len + 1) + t + 1) * g.dim + j] = hn;
      }
      if (unit) s_h[((cur ^ 1) * per_block + sample) * g.dim + j] = hn;
      h = hn;
      cur ^= 1;
      __syncthreads();
    }
    if (live && last) last[b * ldl + j] = h;
    __syncthreads();
  }
}

// dgi: (batch*len, 3E) row (b,t);  dgh: (batch, len+1, 3E) with row 0 zero and row t+1 = dgh_t
__global__ void __launch_bounds__(kBlock)
gru_bwd_kernel(const GruGeom g, const float* __restrict__ gi, int64_t ldgi, const float* __restrict__ w_hh,
               const float* __restrict__ b_hh, const float* __restrict__ hbuf, const float* __restrict__ glast,
               int64_t ldgl, float* __restrict__ dgi, float* __restrict__ dgh) {
  extern __shared__ float lds[];
  const int stride = g.dim + 1;
  const int per_block = kBlock / g.group;
  float* s_w = lds;                                   // [3E][E+1]
  float* s_h = s_w + 3 * g.dim * stride;              // [samples][E]  h_{t-1}
  float* s_d = s_h + per_block * g.dim;               // [samples][3E] dgh_t
  const int sample = threadIdx.x / g.group, j = threadIdx.x % g.group;
  const bool unit = j < g.dim;
  load_whh(s_w, w_hh, g.dim);
  const float br = unit ? b_hh[j] : 0.f, bz = unit ? b_hh[g.dim + j] : 0.f, bn = unit ? b_hh[2 * g.dim + j] : 0.f;
  for (int64_t b0 = (int64_t)blockIdx.x * per_block; b0 < g.batch; b0 += (int64_t)gridDim.x * per_block) {
    const int64_t b = b0 + sample;
    const bool live = unit && b < g.batch;
    float dh = live ? glast[b * ldgl + j] : 0.0f;
    if (live) {
      float* z0 = dgh + b * (g.len + 1) * 3 * g.dim;
      z0[j] = 0.0f; z0[g.dim + j] = 0.0f; z0[2 * g.dim + j] = 0.0f;
    }
    for (int t = g.len - 1; t >= 0; --t) {
      float hp = 0.0f;
      if (live) hp = hbuf[(b * (g.len + 1) + t) * g.dim + j];
      if (unit) s_h[sample * g.dim + j] = hp;
      __syncthreads();
      float dhp = 0.0f;
      if (live) {
        const float* hv = s_h + sample * g.dim;
        float ar = br, az = bz, an = bn;
        const float* wr = s_w + j * stride;
        const float* wz = s_w + (g.dim + j) * stride;
        const float* wn = s_w + (2 * g.dim + j) * stride;
        for (int k = 0; k < g.dim; ++k) {
          const float hk = hv[k];
          ar = fmaf(wr[k], hk, ar);
          az = fmaf(wz[k], hk, az);
          an = fmaf(wn[k], hk, an);
        }
        const float* gir = gi + (b * g.len + t) * ldgi;
        const float r = ctr_sigmoid(gir[j] + ar);
        const float z = ctr_sigmoid(gir[g.dim + j] + az);
        const float n = tanhf(gir[2 * g.dim + j] + r * an);
        const float dz = dh * (hp - n);
        const float dn = dh * (1.0f - z);
        dhp = dh * z;
        const float dan = dn * (1.0f - n * n);
        const float dar = dan * an * r * (1.0f - r);
        const float daz = dz * z * (1.0f - z);
        float* o = dgi + (b * g.len + t) * 3 * g.dim;
        o[j] = dar; o[g.dim + j] = daz; o[2 * g.dim + j] = dan;
        float* q = dgh + (b * (g.len + 1) + t + 1) * 3 * g.dim;
        const float dhn = dan * r;
        q[j] = dar; q[g.dim + j] = daz; q[2 * g.dim + j] = dhn;
        float* sd = s_d + sample * 3 * g.dim;
        sd[j] = dar; sd[g.dim + j] = daz; sd[2 * g.dim + j] = dhn;
      }
      __syncthreads();
      if (live) {
        // dh_{t-1}[j] = dh_t[j]*z + sum_i W_hh[i][j] * dgh_t[i]
        const float* sd = s_d + sample * 3 * g.dim;
        float acc = dhp;
        for (int i = 0; i < 3 * g.dim; ++i) acc = fmaf(s_w[i * stride + j], sd[i], acc);
        dh = acc;
      }
      __syncthreads();
    }
  }
}

// ---- E = 16 (the DIEN config): a sample is one 16-lane DPP row ---------------------------------
// The generic kernels above spend ~240 instructions per lane and step, most of them LDS reads
// of W_hh and of the state plus loop overhead: with 8 waves per SIMD they are instruction-bound
// (330 us forward / 720 us backward for 32768 x 100 steps), not latency-bound (prefetching the
// gate inputs changed nothing).  Here lane j keeps rows j, 16+j, 32+j of W_hh (and, backward, the
// matching columns) in registers and reads h_k / dgh_i of its sample straight out of the
// neighbouring lanes with DPP row_share: no LDS, no barrier, ~3x fewer instructions per step.
template <int K>
__device__ __forceinline__ float row_bcast(float v) {
  // mov_dpp, not update_dpp(old = 0): every lane of a row_share has a source, and an "old" value costs a v_mov per
  // broadcast and keeps the DPP combiner from folding the broadcast into the FMA that uses it
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x150 + K, 0xf, 0xf, true));
}
// acc += w * (lane K of the row's value of src): the broadcast rides on the FMA as a DPP operand (v_fmac_f32_dpp) --
// one instruction instead of a v_mov_b32_dpp and a v_fmac.  The compiler's combiner does not do this for a broadcast
// with three to six users.  The asm is opaque to the hazard recogniser: a DPP source must not have been written by
// the two VALU instructions before it (settle() below).
template <int K>
__device__ __forceinline__ void fmac_bcast(float& acc, float src, float w) {
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(w), "n"(K));
}
// two wait states between the VALU writes of these values and their first DPP read; ties the later asm to them
__device__ __forceinline__ void settle(float& a) { asm volatile("s_nop 1" : "+v"(a)); }
__device__ __forceinline__ void settle(float& a, float& b, float& c, float& d) {
  asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
#define CTR_ROW16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

__global__ void __launch_bounds__(kBlock)
gru16_fwd_kernel(int64_t batch, int len, const float* __restrict__ gi, int64_t ldgi, const float* __restrict__ w_hh,
                 const float* __restrict__ b_hh, float* __restrict__ hbuf, float* __restrict__ last, int64_t ldl) {
  constexpr int E = 16;
  const int j = threadIdx.x & 15;
  float wr[E], wz[E], wn[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    wr[k] = w_hh[j * E + k];
    wz[k] = w_hh[(E + j) * E + k];
    wn[k] = w_hh[(2 * E + j) * E + k];
  }
  const float br = b_hh[j], bz = b_hh[E + j], bn = b_hh[2 * E + j];
  const int64_t stride = (int64_t)gridDim.x * (kBlock / E);
  for (int64_t b = (int64_t)blockIdx.x * (kBlock / E) + (threadIdx.x >> 4); b < batch; b += stride) {
    float h = 0.0f;
    hbuf[b * (len + 1) * E + j] = 0.0f;
    const float* g0 = gi + (b * len) * ldgi;
    float gr = len > 0 ? ctr_ldg(g0 + j) : 0.0f, gz = len > 0 ? ctr_ldg(g0 + E + j) : 0.0f,
          gn = len > 0 ? ctr_ldg(g0 + 2 * E + j) : 0.0f;
    for (int t = 0; t < len; ++t) {
      float nr = 0.0f, nz = 0.0f, nn = 0.0f;
      if (t + 1 < len) {
        const float* g1 = gi + (b * len + t + 1) * ldgi;
        nr = ctr_ldg(g1 + j); nz = ctr_ldg(g1 + E + j); nn = ctr_ldg(g1 + 2 * E + j);
      }
      float ar = br, az = bz, an = bn;
#define CTR_STEP(K) { const float hk = row_bcast<K>(h); ar = fmaf(wr[K], hk, ar); az = fmaf(wz[K], hk, az); an = fmaf(wn[K], hk, an); }
      CTR_ROW16(CTR_STEP)
#undef CTR_STEP
      const float r = ctr_sigmoid(gr + ar);
      const float z = ctr_sigmoid(gz + az);
      const float n = tanhf(gn + r * an);
      h = (1.0f - z) * n + z * h;
      hbuf[(b * (len + 1) + t + 1) * E + j] = h;
      gr = nr; gz = nz; gn = nn;
    }
    if (last) last[b * ldl + j] = h;
  }
}

__global__ void __launch_bounds__(kBlock)
gru16_bwd_kernel(int64_t batch, int len, const float* __restrict__ gi, int64_t ldgi, const float* __restrict__ w_hh,
                 const float* __restrict__ b_hh, const float* __restrict__ hbuf, const float* __restrict__ glast,
                 int64_t ldgl, float* __restrict__ dgi, float* __restrict__ dgh) {
  constexpr int E = 16;
  const int j = threadIdx.x & 15;
  float wr[E], wz[E], wn[E];     // rows j, 16+j, 32+j of W_hh: the forward dots
  float cr[E], cz[E], cn[E];     // column j of the three gate blocks: W_hh^T dgh
#pragma unroll
  for (int k = 0; k < E; ++k) {
    wr[k] = w_hh[j * E + k];
    wz[k] = w_hh[(E + j) * E + k];
    wn[k] = w_hh[(2 * E + j) * E + k];
    cr[k] = w_hh[k * E + j];
    cz[k] = w_hh[(E + k) * E + j];
    cn[k] = w_hh[(2 * E + k) * E + j];
  }
  const float br = b_hh[j], bz = b_hh[E + j], bn = b_hh[2 * E + j];
  const int64_t stride = (int64_t)gridDim.x * (kBlock / E);
  for (int64_t b = (int64_t)blockIdx.x * (kBlock / E) + (threadIdx.x >> 4); b < batch; b += stride) {
    float dh = glast[b * ldgl + j];
    float* z0 = dgh + b * (len + 1) * 3 * E;
    z0[j] = 0.0f; z0[E + j] = 0.0f; z0[2 * E + j] = 0.0f;
    float hp_next = 0.0f, gr = 0.0f, gz = 0.0f, gn = 0.0f;
    if (len > 0) {
      hp_next = ctr_ldg(hbuf + (b * (len + 1) + len - 1) * E + j);
      const float* g0 = gi + (b * len + len - 1) * ldgi;
      gr = ctr_ldg(g0 + j); gz = ctr_ldg(g0 + E + j); gn = ctr_ldg(g0 + 2 * E + j);
    }
    for (int t = len - 1; t >= 0; --t) {
      const float hp = hp_next;
      float nr = 0.0f, nz = 0.0f, nn = 0.0f;
      if (t > 0) {
        hp_next = ctr_ldg(hbuf + (b * (len + 1) + t - 1) * E + j);
        const float* g1 = gi + (b * len + t - 1) * ldgi;
        nr = ctr_ldg(g1 + j); nz = ctr_ldg(g1 + E + j); nn = ctr_ldg(g1 + 2 * E + j);
      }
      float ar = br, az = bz, an = bn;
#define CTR_STEP(K) { const float hk = row_bcast<K>(hp); ar = fmaf(wr[K], hk, ar); az = fmaf(wz[K], hk, az); an = fmaf(wn[K], hk, an); }
      CTR_ROW16(CTR_STEP)
#undef CTR_STEP
      const float r = ctr_sigmoid(gr + ar);
      const float z = ctr_sigmoid(gz + az);
      const float n = tanhf(gn + r * an);
      const float dz = dh * (hp - n);
      const float dn = dh * (1.0f - z);
      const float dan = dn * (1.0f - n * n);
      const float dar = dan * an * r * (1.0f - r);
      const float daz = dz * z * (1.0f - z);
      const float dhn = dan * r;
      float* o = dgi + (b * len + t) * 3 * E;
      o[j] = dar; o[E + j] = daz; o[2 * E + j] = dan;
      float* q = dgh + (b * (len + 1) + t + 1) * 3 * E;
      q[j] = dar; q[E + j] = daz; q[2 * E + j] = dhn;
      // dh_{t-1}[j] = dh_t[j]*z + sum_i W_hh[i][j] * dgh_t[i]
      float acc = dh * z;
#define CTR_STEP(K) { acc = fmaf(cr[K], row_bcast<K>(dar), acc); acc = fmaf(cz[K], row_bcast<K>(daz), acc); acc = fmaf(cn[K], row_bcast<K>(dhn), acc); }
      CTR_ROW16(CTR_STEP)
#undef CTR_STEP
      dh = acc;
      gr = nr; gz = nz; gn = nn;
    }
  }
}

// ---- E = 16 with the input projection inside (ctr_gru_fused_fwd / _bwd): the matrix-core kernels further down.
// gi = X W_ih^T + b_ih as a GEMM writes and re-reads 3E floats per step and sample (629 MB each way at the DIEN
// config), and its backward is three more passes (dX, dW_ih, dW_hh) over dgi / dgh of the same size; the fused
// kernels store nothing of size 3E per step.  One partial per workgroup goes to the workspace (reduce_segments).
// (Round 2 also had a DPP-row form of the fused kernels -- four samples per wave, 775 us against 365 us backward;
// retired in round 3.  The independent cross-check of the fused path is the GEMM decomposition above.)
constexpr int kGruSlab = 2 * 48 * 16 + 2 * 48;   // dW_ih, dW_hh, db_ih, db_hh

typedef float gru_f32x4 __attribute__((ext_vector_type(4)));

// ---- E = 16 on the matrix cores: sixteen samples per wave ------------------------------------------------------------
// The DPP kernels above spend a step's time issuing 96-192 broadcast FMAs for FOUR samples.  As a matrix product the
// same work is  P_g (16 units x 16 samples) = W_g (16 x 16) . V (16 units x 16 samples)  -- four v_mfma_f32_16x16x4_f32
// per gate block, 32 cycles each, for SIXTEEN samples.  What makes it a recurrence without any data movement: the
// result layout of that instruction (lane (q, n) holds units 4q .. 4q+3 of sample n) is also a legal B-operand layout
// for the next step if contraction chunk c is taken to be the units {4q + c}: the state registers of a lane ARE its B
// operands, and W enters as A operands  W_g[lane % 16][4q + c]  (12 registers per matrix).  x_t arrives as one
// dwordx4 per lane (units 4q .. 4q+3 of its sample) and is a B operand the same way.
// Gate functions for the matrix-core kernels: a lane evaluates them for four units per step, and expf + IEEE division +
// ocml's branching tanhf were ~90 instructions per unit (400 of the 500 VALU instructions of a step).  v_exp_f32 and
// v_rcp_f32 are accurate to 1 ulp of their own operation; through the argument scaling the results carry an absolute
// error of a few 1e-7, two orders inside the 1e-5 the parity tests allow on states and gradients.
__device__ __forceinline__ float gate_sigmoid(float z) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * z));
}
__device__ __forceinline__ float gate_tanh(float z) {   // 1 - 2 / (1 + e^{2z}); saturates cleanly (rcp(inf) = 0)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681472f * z));
}

__global__ void __launch_bounds__(kBlock)
gru16_mfma_fwd_kernel(int64_t batch, int len, const float* __restrict__ x, int64_t ldx, const float* __restrict__ w_ih,
                      const float* __restrict__ b_ih, const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                      float* __restrict__ hbuf, float* __restrict__ last, int64_t ldl) {
  constexpr int E = 16;
  const int lane = threadIdx.x & 63, q = lane >> 4, lo = lane & 15;
  float ah[3][4], ai[3][4];     // A operands: W_g[unit lo][input 4q + c]
  gru_f32x4 ch[3], ci[3];       // biases in accumulator layout: unit 4q + r
#pragma unroll
  for (int g = 0; g < 3; ++g) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ah[g][c] = w_hh[(g * E + lo) * E + 4 * q + c];
      ai[g][c] = w_ih[(g * E + lo) * E + 4 * q + c];
      ch[g][c] = b_hh[g * E + 4 * q + c];
      ci[g][c] = b_ih[g * E + 4 * q + c];
    }
  }
  const int64_t waves = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t s0 = ((((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6)) * 16; s0 < batch; s0 += waves * 16) {
    const int64_t smp = s0 + lo;
    const bool live = smp < batch;
    const int64_t sc = live ? smp : batch - 1;          // dead lanes compute on a copy of the last sample, store nothing
    const float* xs = x + (sc * len) * ldx + 4 * q;
    float* hs = hbuf + (sc * (len + 1)) * E + 4 * q;
    gru_f32x4 h = {0.f, 0.f, 0.f, 0.f};
    if (live) *reinterpret_cast<gru_f32x4*>(hs) = h;
    // Memory traffic in blocks of four steps, ordered so that nothing is ever waited for right after it was issued
    // (loads and stores share one in-order counter; PMC showed 25 % of the wave's life in s_waitcnt with the states
    // stored step by step): at the end of a block first take over the x values requested a block ago, then store
    // the block's four states, then request x of the block after next.
    const gru_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    gru_f32x4 xc[4], xq[4], hout[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xc[u] = u < len ? *reinterpret_cast<const gru_f32x4*>(xs + (int64_t)u * ldx) : zero4;
#pragma unroll
    for (int u = 0; u < 4; ++u) xq[u] = 4 + u < len ? *reinterpret_cast<const gru_f32x4*>(xs + (int64_t)(4 + u) * ldx) : zero4;
    for (int t0 = 0; t0 < len; t0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u;
        hout[u] = h;
        if (t >= len) continue;
        gru_f32x4 gi[3], gh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          gi[g] = ci[g];
          gh[g] = ch[g];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            gi[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[g][c], xc[u][c], gi[g], 0, 0, 0);
            gh[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[g][c], h[c], gh[g], 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float rg = gate_sigmoid(gi[0][r] + gh[0][r]);
          const float zg = gate_sigmoid(gi[1][r] + gh[1][r]);
          const float ng = gate_tanh(gi[2][r] + rg * gh[2][r]);
          h[r] = (1.0f - zg) * ng + zg * h[r];
        }
        hout[u] = h;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) xc[u] = xq[u];
      if (live) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (t0 + u < len) *reinterpret_cast<gru_f32x4*>(hs + (int64_t)(t0 + u + 1) * E) = hout[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        xq[u] = t0 + 8 + u < len ? *reinterpret_cast<const gru_f32x4*>(xs + (int64_t)(t0 + 8 + u) * ldx) : zero4;
    }
    if (last && live) {
#pragma unroll
      for (int r = 0; r < 4; ++r) last[smp * ldl + 4 * q + r] = h[r];
    }
  }
}

// Backward on the matrix cores, same ownership (lane (q, n): units 4q .. 4q+3 of sample n, sixteen samples per wave).
// Per step: the forward products again (24 MFMAs); W_hh^T dgh and W_ih^T dgi with the gate gradients as B operands
// straight out of their registers (24; the results land in the layout of dh and of the dX row to store); and the
// weight gradients  dW_g[i][k] += sum_n dG_g[i][n] V[k][n]  -- a contraction over the SAMPLES, for which both operands
// are needed unit-major (lane (q, u): unit u of samples 4q .. 4q+3): x_t / h_{t-1} are simply loaded a second time
// in that order, the four gate-gradient tiles are transposed through a 5 KB strip of LDS private to the wave (24).
// 72 matrix instructions of 32 cycles per step for sixteen samples, where the DPP kernel issues ~300 VALU
// instructions, half of them at DPP rate, for four.
constexpr int kGruTS = 20;   // row stride of a transposed tile: 16-byte aligned rows, two-way write conflicts at most

__global__ void __launch_bounds__(kBlock)
gru16_mfma_bwd_kernel(int64_t batch, int len, const float* __restrict__ x, int64_t ldx, const float* __restrict__ w_ih,
                      const float* __restrict__ b_ih, const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                      const float* __restrict__ hbuf, const float* __restrict__ glast, int64_t ldgl,
                      float* __restrict__ gx, int64_t ldgx, float* __restrict__ ws) {
  constexpr int E = 16;
  __shared__ __attribute__((aligned(16))) float s_t[kBlock / 64][4][E * kGruTS];   // per wave: dar, daz, dan, dhn tiles
  __shared__ float s_part[kBlock / 64][kGruSlab];
  const int lane = threadIdx.x & 63, q = lane >> 4, lo = lane & 15, wave = threadIdx.x >> 6;
  float ah[3][4], ai[3][4];     // forward A operands: W_g[unit lo][input 4q + c]
  float th[3][4], ti[3][4];     // transposed A operands: W_g[unit 4q + c][input lo]
  gru_f32x4 ch[3], ci[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ah[g][c] = w_hh[(g * E + lo) * E + 4 * q + c];
      ai[g][c] = w_ih[(g * E + lo) * E + 4 * q + c];
      th[g][c] = w_hh[(g * E + 4 * q + c) * E + lo];
      ti[g][c] = w_ih[(g * E + 4 * q + c) * E + lo];
      ch[g][c] = b_hh[g * E + 4 * q + c];
      ci[g][c] = b_ih[g * E + 4 * q + c];
    }
  }
  const gru_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  gru_f32x4 mi[3], mh[3];      // dW_ih / dW_hh gate blocks: register r = row 4q + r, column lo
  gru_f32x4 sb[4];             // sums of dar, daz, dan, dhn over this lane's (sample, steps): units 4q + r
#pragma unroll
  for (int g = 0; g < 3; ++g) mi[g] = mh[g] = zero4;
#pragma unroll
  for (int g = 0; g < 4; ++g) sb[g] = zero4;
  float* tile = s_t[wave][0];
  const int64_t waves = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t s0 = ((((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6)) * 16; s0 < batch; s0 += waves * 16) {
    const int64_t smp = s0 + lo;
    const bool live = smp < batch;
    const int64_t sc = live ? smp : batch - 1;
    // a dead lane starts from a zero gradient: every gate gradient it forms is zero, it adds nothing anywhere
    gru_f32x4 dh = zero4;
    if (live) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[r] = glast[smp * ldgl + 4 * q + r];
    }
    const float* xs = x + (sc * len) * ldx + 4 * q;                 // sample-major: units 4q .. 4q+3 of sample lo
    const float* hs = hbuf + (sc * (len + 1)) * E + 4 * q;
    // unit-major: unit lo of samples 4q + c (clamped like the others; dead samples meet zero gate gradients)
    const float* xts[4];
    const float* hts[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int64_t sq = s0 + 4 * q + c;
      if (sq >= batch) sq = batch - 1;
      xts[c] = x + (sq * len) * ldx + lo;
      hts[c] = hbuf + (sq * (len + 1)) * E + lo;
    }
    gru_f32x4 xv = zero4, hv = zero4, xt = zero4, ht = zero4;
    if (len > 0) {
      const int64_t t = len - 1;
      xv = *reinterpret_cast<const gru_f32x4*>(xs + t * ldx);
      hv = *reinterpret_cast<const gru_f32x4*>(hs + t * E);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        xt[c] = ctr_ldg(xts[c] + t * ldx);
        ht[c] = ctr_ldg(hts[c] + t * E);
      }
    }
    for (int t = len - 1; t >= 0; --t) {
      // step t-1's operands, requested a step ahead
      gru_f32x4 xvn = zero4, hvn = zero4, xtn = zero4, htn = zero4;
      if (t > 0) {
        const int64_t tp = t - 1;
        xvn = *reinterpret_cast<const gru_f32x4*>(xs + tp * ldx);
        hvn = *reinterpret_cast<const gru_f32x4*>(hs + tp * E);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          xtn[c] = ctr_ldg(xts[c] + tp * ldx);
          htn[c] = ctr_ldg(hts[c] + tp * E);
        }
      }
      gru_f32x4 gi[3], gh[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        gi[g] = ci[g];
        gh[g] = ch[g];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          gi[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[g][c], xv[c], gi[g], 0, 0, 0);
          gh[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[g][c], hv[c], gh[g], 0, 0, 0);
        }
      }
      gru_f32x4 dar, daz, dan, dhn, zz;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float rg = gate_sigmoid(gi[0][r] + gh[0][r]);
        const float zg = gate_sigmoid(gi[1][r] + gh[1][r]);
        const float an = gh[2][r];
        const float ng = gate_tanh(gi[2][r] + rg * an);
        const float dz = dh[r] * (hv[r] - ng);
        const float dn = dh[r] * (1.0f - zg);
        dan[r] = dn * (1.0f - ng * ng);
        dar[r] = dan[r] * an * rg * (1.0f - rg);
        daz[r] = dz * zg * (1.0f - zg);
        dhn[r] = dan[r] * rg;
        zz[r] = zg;
      }
      sb[0] += dar; sb[1] += daz; sb[2] += dan; sb[3] += dhn;
      // W^T d with the gate gradients as B operands as they are: dh_{t-1} - dh_t z and the dX row
      gru_f32x4 tacc = zero4, xacc = zero4;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        tacc = __builtin_amdgcn_mfma_f32_16x16x4f32(th[0][c], dar[c], tacc, 0, 0, 0);
        xacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ti[0][c], dar[c], xacc, 0, 0, 0);
        tacc = __builtin_amdgcn_mfma_f32_16x16x4f32(th[1][c], daz[c], tacc, 0, 0, 0);
        xacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ti[1][c], daz[c], xacc, 0, 0, 0);
        tacc = __builtin_amdgcn_mfma_f32_16x16x4f32(th[2][c], dhn[c], tacc, 0, 0, 0);
        xacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ti[2][c], dan[c], xacc, 0, 0, 0);
      }
      // the four gate-gradient tiles unit-major: tile[unit][sample]
      {
        const gru_f32x4* src[4] = {&dar, &daz, &dan, &dhn};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r) tile[k * (E * kGruTS) + (4 * q + r) * kGruTS + lo] = (*src[k])[r];
      }
      __builtin_amdgcn_wave_barrier();
      gru_f32x4 dT[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) dT[k] = *reinterpret_cast<const gru_f32x4*>(tile + k * (E * kGruTS) + lo * kGruTS + 4 * q);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        mi[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT[0][c], xt[c], mi[0], 0, 0, 0);
        mi[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT[1][c], xt[c], mi[1], 0, 0, 0);
        mi[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT[2][c], xt[c], mi[2], 0, 0, 0);
        mh[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT[0][c], ht[c], mh[0], 0, 0, 0);
        mh[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT[1][c], ht[c], mh[1], 0, 0, 0);
        mh[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT[3][c], ht[c], mh[2], 0, 0, 0);
      }
      if (live) *reinterpret_cast<gru_f32x4*>(gx + (smp * len + t) * ldgx + 4 * q) = xacc;
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[r] = fmaf(dh[r], zz[r], tacc[r]);
      xv = xvn; hv = hvn; xt = xtn; ht = htn;
    }
  }
  // wave partial -> LDS slab [dW_ih 48x16 | dW_hh 48x16 | db_ih 48 | db_hh 48], the four waves summed in wave order
  float* mine = s_part[wave];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mine[(g * E + 4 * q + r) * E + lo] = mi[g][r];
      mine[768 + (g * E + 4 * q + r) * E + lo] = mh[g][r];
    }
  // bias sums: over the sixteen samples of the wave (the lanes of a DPP row)
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = sb[k][r];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
      sb[k][r] = v;
    }
  if (lo == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mine[1536 + 0 * E + 4 * q + r] = sb[0][r]; mine[1536 + 1 * E + 4 * q + r] = sb[1][r]; mine[1536 + 2 * E + 4 * q + r] = sb[2][r];
      mine[1584 + 0 * E + 4 * q + r] = sb[0][r]; mine[1584 + 1 * E + 4 * q + r] = sb[1][r]; mine[1584 + 2 * E + 4 * q + r] = sb[3][r];
    }
  }
  __syncthreads();
  float* out = ws + (int64_t)blockIdx.x * kGruSlab;
  for (int i = threadIdx.x; i < kGruSlab; i += kBlock)
    out[i] = (s_part[0][i] + s_part[1][i]) + (s_part[2][i] + s_part[3][i]);
}

inline size_t fwd_lds(const GruGeom& g) {
  return sizeof(float) * (3 * g.dim * (g.dim + 1) + 2 * (kBlock / g.group) * g.dim);
}
inline size_t bwd_lds(const GruGeom& g) {
  return sizeof(float) * (3 * g.dim * (g.dim + 1) + 4 * (kBlock / g.group) * g.dim);
}

inline int make_geom(int64_t batch, int len, int dim, GruGeom* g) {
  CTR_REQUIRE(dim >= 1 && dim <= 64, CTR_ELIMIT);  // W_hh must fit LDS, one lane per hidden unit
  g->batch = batch;
  g->len = len;
  g->dim = dim;
  g->group = pow2_ceil(dim);
  return CTR_OK;
}

inline int grid_for(const GruGeom& g) {
  const int64_t blocks = ctr_ceil_div(g.batch, kBlock / g.group);
  return (int)(blocks < 2048 ? blocks : 2048);
}

}  // namespace

extern "C" int ctr_gru_fwd(const float* gi, int64_t ldgi, const float* w_hh, const float* b_hh, int64_t batch, int len,
                           int dim, float* hbuf, float* last, int64_t ldl, void* stream) {
  CTR_REQUIRE(batch >= 0 && len >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(w_hh && b_hh && hbuf && (len == 0 || gi) && ldgi >= 3 * (int64_t)dim, CTR_EINVAL);
  CTR_REQUIRE(!last || ldl >= dim, CTR_EINVAL);
  GruGeom g;
  int rc = make_geom(batch, len, dim, &g);
  if (rc != CTR_OK) return rc;
  if (dim == 16 && batch % 4 == 0) {  // whole waves of live 16-lane rows (DPP reads cross lanes of a row only)
    hipLaunchKernelGGL(gru16_fwd_kernel, dim3(grid_for(g)), dim3(kBlock), 0, (hipStream_t)stream, batch, len, gi, ldgi,
                       w_hh, b_hh, hbuf, last, ldl);
    return ctr_launch_status();
  }
  hipLaunchKernelGGL(gru_fwd_kernel, dim3(grid_for(g)), dim3(kBlock), fwd_lds(g), (hipStream_t)stream, g, gi, ldgi, w_hh,
                     b_hh, hbuf, last, ldl);
  return ctr_launch_status();
}

extern "C" int ctr_gru_bwd(const float* gi, int64_t ldgi, const float* w_hh, const float* b_hh, const float* hbuf,
                           int64_t batch, int len, int dim, const float* glast, int64_t ldgl, float* dgi, float* dgh,
                           void* stream) {
  CTR_REQUIRE(batch >= 0 && len >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(w_hh && b_hh && hbuf && glast && dgh && (len == 0 || (gi && dgi)), CTR_EINVAL);
  CTR_REQUIRE(ldgi >= 3 * (int64_t)dim && ldgl >= dim, CTR_EINVAL);
  GruGeom g;
  int rc = make_geom(batch, len, dim, &g);
  if (rc != CTR_OK) return rc;
  if (dim == 16 && batch % 4 == 0) {
    hipLaunchKernelGGL(gru16_bwd_kernel, dim3(grid_for(g)), dim3(kBlock), 0, (hipStream_t)stream, batch, len, gi, ldgi,
                       w_hh, b_hh, hbuf, glast, ldgl, dgi, dgh);
    return ctr_launch_status();
  }
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(grid_for(g)), dim3(kBlock), bwd_lds(g), (hipStream_t)stream, g, gi, ldgi, w_hh,
                     b_hh, hbuf, glast, ldgl, dgi, dgh);
  return ctr_launch_status();
}

// C ABI (include/ctrhip.h): the GRU with its input projection inside.  dim == 16, 16-byte aligned rows only (CTR_ELIMIT
// otherwise, nothing enqueued: the caller forms gi with ctr_linear_fwd and uses ctr_gru_fwd / ctr_gru_bwd).
extern "C" int ctr_gru_fused_fwd(const float* x, int64_t ldx, const float* w_ih, const float* b_ih, const float* w_hh,
                                 const float* b_hh, int64_t batch, int len, int dim, float* hbuf, float* last,
                                 int64_t ldl, void* stream) {
  CTR_REQUIRE(batch >= 0 && len >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(w_ih && b_ih && w_hh && b_hh && hbuf && (len == 0 || x) && ldx >= dim, CTR_EINVAL);
  CTR_REQUIRE(!last || ldl >= dim, CTR_EINVAL);
  CTR_REQUIRE(dim == 16, CTR_ELIMIT);
  // sixteen samples per wave on the matrix cores; any batch (a partial wave repeats its last sample)
  CTR_REQUIRE(ldx % 4 == 0 && ctr_aligned16(x) && ctr_aligned16(hbuf), CTR_ELIMIT);
  int64_t grid = ctr_ceil_div(batch, (kBlock / 64) * 16);
  if (grid > 256 * 4) grid = 256 * 4;
  hipLaunchKernelGGL(gru16_mfma_fwd_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, batch, len, x,
                     ldx, w_ih, b_ih, w_hh, b_hh, hbuf, last, ldl);
  return ctr_launch_status();
}

extern "C" int ctr_gru_fused_bwd(const float* x, int64_t ldx, const float* w_ih, const float* b_ih, const float* w_hh,
                                 const float* b_hh, const float* hbuf, int64_t batch, int len, int dim,
                                 const float* glast, int64_t ldgl, float* gx, int64_t ldgx, float* gw_ih, float* gb_ih,
                                 float* gw_hh, float* gb_hh, float* workspace, int64_t workspace_floats, void* stream) {
  CTR_REQUIRE(batch >= 0 && len >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(w_ih && b_ih && w_hh && b_hh && hbuf && glast && gw_ih && gb_ih && gw_hh && gb_hh, CTR_EINVAL);
  CTR_REQUIRE((len == 0 || (x && gx)) && ldx >= dim && ldgx >= dim && ldgl >= dim, CTR_EINVAL);
  CTR_REQUIRE(dim == 16, CTR_ELIMIT);
  if (len == 0) return CTR_OK;
  CTR_REQUIRE(ldx % 4 == 0 && ldgx % 4 == 0 && ctr_aligned16(x) && ctr_aligned16(hbuf) && ctr_aligned16(gx), CTR_ELIMIT);
  int64_t grid = ctr_ceil_div(batch, (kBlock / 64) * 16);   // sixteen samples per wave
  if (grid > 256 * 2) grid = 256 * 2;
  CTR_REQUIRE(workspace && workspace_floats >= grid * kGruSlab, CTR_EINVAL);
  hipLaunchKernelGGL(gru16_mfma_bwd_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, batch, len, x,
                     ldx, w_ih, b_ih, w_hh, b_hh, hbuf, glast, ldgl, gx, ldgx, workspace);
  int rc = ctr_launch_status();
  if (rc != CTR_OK) return rc;
  CtrSegments segs;
  segs.n = 4;
  segs.s[0] = CtrSegment{0, 768, gw_ih};
  segs.s[1] = CtrSegment{768, 768, gw_hh};
  segs.s[2] = CtrSegment{1536, 48, gb_ih};
  segs.s[3] = CtrSegment{1584, 48, gb_hh};
  return ctr_reduce_segments(workspace, (int)grid, kGruSlab, segs, (hipStream_t)stream);
}
