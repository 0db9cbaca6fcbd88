// Input gradient  gX[m, :k] (=|+=) sum_n gZ[m, n] W[n, :k],  gZ = gY * act'(Y),  with gY, Y and W copied
// global -> LDS directly (global_load_lds_dwordx4) through a three-stage ring; see gemm_dlds.hip for the
// ring and gemm_dlds_dw.hip for why the loads are issued from asm.
//
// Tile 128 rows x 32*NT input columns per 256-thread workgroup, persistent over the row tiles; a step
// contracts 16 units.  gY / Y tiles are [128][16] with the chunk swizzle of gemm_dlds.hip (read with
// ds_read_b128 along the units), the W tile is the plain row-major [16][32*NT] block and is read down
// its rows (32 consecutive floats per half-wave: conflict-free).  act'(Y) is applied as gY is read.
// The 16-byte direct loads only need 4-byte aligned addresses, so any layout with n, k >= 4 is taken.
// No load leaves its matrix: a chunk that would cross the end of a row is fetched from 4 floats before
// the end instead.  Along the units (contraction tail, n % 16 != 0) the gY fragment zeroes the
// positions that are then duplicates or past n; along W's columns (k % 4 != 0) the reader adds the
// shift to its column index.
#include "ctr_common.h"
#include <stdlib.h>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kBM = 128;
constexpr int kBK = 16;
constexpr int kStages = 3;

struct DxArgs {
  const float* gy; int64_t ldgy;
  const float* y; int64_t ldy;
  const float* w; int64_t ldw;
  float* gx; int64_t ldgx;
  int64_t m; int n; int k; int accumulate;
  // optional epilogue (EPI instantiations; DIN attention, see ctr_linear_dx_masked): gx *= act'(xin) -- the layer's
  // own input is the previous layer's activation output -- and per-group column sums of the masked gx added to
  // gsum[row / group]
  const float* xin; int64_t ldxin; int act_in;
  float* gsum; int64_t ldgsum; int group;
  // instead of xin: the sign bits ctr_linear_group_fwd wrote (bit (c & 31) of xmask[i*ldxmask + c/32]); gx *= bit
  const uint32_t* xmask; int64_t ldxmask;
  // EPI == 2 (ctr_linear_dx_scatter): gx is not stored; row i (+ attn[i] * gpool[i / group, :]) is ADDED to row
  // idx[i] of a (vocab, k) table gradient -- DIN's history gradient without the (B*L, E) intermediate
  const int64_t* sc_idx; const float* sc_attn; const float* sc_gpool; int64_t sc_ldgp; float* sc_table; int64_t sc_vocab;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}

// (m0 is a reserved register: the compiler only sets it right in front of an instruction that reads it,
// never keeps a value there, and rejects it as a clobber)
__device__ __forceinline__ void dma16(const float* g, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_base) : "memory");
}

__device__ __forceinline__ uint32_t lds_addr(const float* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

// [128 rows][16 units] tile, slot q (16 B) = row q/4, unit chunk (q & 3) ^ ((row >> 1) & 3)
__device__ __forceinline__ void fetch_rows(float* stage, const float* __restrict__ src, int64_t ld, int64_t row0,
                                           int64_t rows_total, int u0, int u_total, int lane, int wave) {
#pragma unroll
  for (int i = 0; i < kBM * 4 / kThreads; ++i) {
    const int q0 = 64 * wave + kThreads * i;
    const int q = q0 + lane;
    const int row = q >> 2, c = (q & 3) ^ ((row >> 1) & 3);
    int64_t gr = row0 + row;
    gr = gr < rows_total ? gr : rows_total - 1;
    int u = u0 + c * 4;
    u = u < u_total - 4 ? u : u_total - 4;
    dma16(src + gr * ld + u, __builtin_amdgcn_readfirstlane(lds_addr(stage + q0 * 4)));
  }
}

// logical column `col` of a matrix with `total` columns sits this many floats further right in its
// (shifted) last chunk
__device__ __forceinline__ int tail_shift(int col, int total) {
  return (total & 3) && col >= (total & ~3) && col < total ? 4 - (total & 3) : 0;
}

// plain [16 units][W columns] block of the weight
template <int W>
__device__ __forceinline__ void fetch_w(float* stage, const float* __restrict__ w, int64_t ldw, int u0, int u_total,
                                        int col0, int cols_total, int lane, int wave) {
  constexpr int kPerRow = W / 4, kChunks = 16 * kPerRow;
  constexpr int kIters = (kChunks + kThreads - 1) / kThreads;
#pragma unroll
  for (int i = 0; i < kIters; ++i) {
    int q0 = 64 * wave + kThreads * i;
    if (kChunks % kThreads != 0 && q0 >= kChunks) q0 -= kChunks;  // narrow tile: fetched twice, same bytes
    const int q = q0 + lane;
    const int row = q / kPerRow, cc = q % kPerRow;
    int col = col0 + cc * 4;
    col = col < cols_total - 4 ? col : cols_total - 4;
    // row = 4*chunk + j of the step follows the unit the gY fragment holds at that position
    // (fetch_rows: chunks crossing n start at n-4)
    int u = u0 + (row & ~3);
    u = (u < u_total - 4 ? u : u_total - 4) + (row & 3);
    dma16(w + (int64_t)u * ldw + col, __builtin_amdgcn_readfirstlane(lds_addr(stage + q0 * 4)));
  }
}

template <int NT, int ACT, int EPI = 0>   // EPI: 0 plain, 1 masked write-back + group sums, 2 scatter-add by row ids
__global__ void __launch_bounds__(kThreads, 2)
gemm_dx_dlds_kernel(const DxArgs a) {
  constexpr int BW = 32 * NT;
  constexpr bool has_y = ACT != CTR_ACT_NONE;
  constexpr int kLoadsW = (16 * (BW / 4) + kThreads - 1) / kThreads;
  constexpr int kPerSlot = (has_y ? 2 : 1) * (kBM * 4 / kThreads) + kLoadsW;
  __shared__ __attribute__((aligned(16))) float s_gy[kStages][kBM * kBK];
  __shared__ __attribute__((aligned(16))) float s_y[has_y ? kStages : 1][has_y ? kBM * kBK : 4];
  __shared__ __attribute__((aligned(16))) float s_w[kStages][kBK * BW];

  const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t mtiles = (a.m + kBM - 1) / kBM;
  const int c0 = blockIdx.y * BW;
  const int nk = (a.n + kBK - 1) / kBK;
  const bool utail = a.n % kBK != 0;
  int64_t tile = blockIdx.x;
  if (tile >= mtiles) return;

  constexpr int CH = 4 / NT;
  floatx16 acc[NT][CH];
  // EPI 2: what this lane's columns add to the padding row (id 0 is a quarter of DIN's history positions: one
  // atomic per sample there is a chain of same-address adds), flushed once at the end
  float padacc[NT];
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) padacc[nb] = 0.0f;

  auto advance = [&](int64_t& t, int& ks) {
    if (++ks == nk) {
      ks = 0;
      t += gridDim.x;
    }
  };
  auto issue = [&](int stage, int64_t t, int ks) {
    fetch_rows(s_gy[stage], a.gy, a.ldgy, t * kBM, a.m, ks * kBK, a.n, lane0, wave);
    if (has_y) fetch_rows(s_y[stage], a.y, a.ldy, t * kBM, a.m, ks * kBK, a.n, lane0, wave);
    fetch_w<BW>(s_w[stage], a.w, a.ldw, ks * kBK, a.n, c0, a.k, lane0, wave);
  };
  int64_t t1 = tile, t2;
  int k1 = 0, k2;
  advance(t1, k1);
  t2 = t1;
  k2 = k1;
  advance(t2, k2);
  issue(0, tile, 0);
  if (t1 < mtiles) issue(1, t1, k1);
  int stage = 0;
  for (; tile < mtiles; tile += gridDim.x) {
    const int64_t i0 = tile * kBM;
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int r = lane & 31, h = lane >> 5;
    const int arow = 32 * wave + r, sw = (arow >> 1) & 3;
    int wcol[NT];
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) wcol[nb] = 32 * nb + r + tail_shift(c0 + 32 * nb + r, a.k);
#pragma unroll
    for (int nb = 0; nb < NT; ++nb)
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nb][c][e] = 0.0f;
    // sign-bit words of this wave's 32 rows (lane r: row r, one word per column tile), requested in front of the
    // contraction: a load issued in the epilogue is waited for together with the stores of the tiles before it
    int idrow = -1;
    float attnrow = 0.0f, sg0[NT], sg1[NT];
    int64_t sc_edge = 0;
    if constexpr (EPI == 2) {
      // lane r: id and attention weight of row r of the wave's 32; per column tile the pooled gradient of the (at
      // most two, group >= 32) samples those rows belong to -- all requested in front of the contraction
      const int64_t first = i0 + 32 * wave;
      const int64_t row = first + r;
      if (row < a.m) {
        const int64_t t = ctr_ldg(a.sc_idx + row);
        idrow = (t >= 0 && t < a.sc_vocab) ? (int)t : -1;
        attnrow = ctr_ldg(a.sc_attn + row);
      }
      const int64_t b0 = (first < a.m ? first : a.m - 1) / a.group;
      sc_edge = (b0 + 1) * (int64_t)a.group;
      const int64_t b1 = sc_edge < a.m ? b0 + 1 : b0;
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) {
        const int col = c0 + 32 * nb + r;
        sg0[nb] = ctr_ldg(a.sc_gpool + b0 * a.sc_ldgp + col);
        sg1[nb] = ctr_ldg(a.sc_gpool + b1 * a.sc_ldgp + col);
      }
    }
    uint32_t mrow[NT];
    if constexpr (EPI == 1) {
      if (a.xmask) {
        const int64_t row = i0 + 32 * wave + r;
#pragma unroll
        for (int nb = 0; nb < NT; ++nb)
          mrow[nb] = row < a.m ? __builtin_nontemporal_load(a.xmask + row * a.ldxmask + ((c0 >> 5) + nb)) : 0u;
      }
    }
    for (int ks = 0; ks < nk; ++ks) {
      if (t1 < mtiles) wait_vmcnt<kPerSlot>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int refill = stage + 2;
      refill = refill >= kStages ? refill - kStages : refill;
      if (t2 < mtiles) issue(refill, t2, k2);
      advance(t1, k1);
      advance(t2, k2);
      float fa[8];
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int slot = arow * 4 + ((2 * h + v) ^ sw);
        const float4 g = *reinterpret_cast<const float4*>(&s_gy[stage][slot * 4]);
        fa[4 * v + 0] = g.x; fa[4 * v + 1] = g.y; fa[4 * v + 2] = g.z; fa[4 * v + 3] = g.w;
        if (has_y) {
          const float4 yv = *reinterpret_cast<const float4*>(&s_y[stage][slot * 4]);
          fa[4 * v + 0] *= ctr_act_grad(yv.x, ACT);
          fa[4 * v + 1] *= ctr_act_grad(yv.y, ACT);
          fa[4 * v + 2] *= ctr_act_grad(yv.z, ACT);
          fa[4 * v + 3] *= ctr_act_grad(yv.w, ACT);
        }
        if (utail && ks == nk - 1) {
          // chunk (2h + v) of the last step was fetched from n-4 if it crosses n: its first `over`
          // positions repeat the previous chunk (or all of it is past n)
          const int over = ks * kBK + (2 * h + v) * 4 + 4 - a.n;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < over) fa[4 * v + j] = 0.0f;
        }
      }
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) {
        float fb[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) fb[t] = s_w[stage][(8 * h + t) * BW + wcol[nb]];
#pragma unroll
        for (int t = 0; t < 8; ++t)
          acc[nb][t % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc[nb][t % CH], 0, 0, 0);
      }
      stage = stage + 1 == kStages ? 0 : stage + 1;
    }
    // C/D map: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if constexpr (EPI == 2) {
      const int64_t first = i0 + 32 * wave;
#pragma unroll
      for (int nb = 0; nb < NT; ++nb) {
        const int col = c0 + 32 * nb + r;   // k % 32 == 0 (host): every column exists
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[nb][0][e];
#pragma unroll
          for (int c = 1; c < CH; ++c) v += acc[nb][c][e];
          const int rho = (e & 3) + 8 * (e >> 2);
          const int id0 = __builtin_amdgcn_readlane(idrow, rho), id1 = __builtin_amdgcn_readlane(idrow, rho + 4);
          const float at0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, attnrow), rho));
          const float at1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, attnrow), rho + 4));
          const int id = h ? id1 : id0;
          const int64_t i = first + rho + 4 * h;
          v = fmaf(h ? at1 : at0, i < sc_edge ? sg0[nb] : sg1[nb], v);
          if (id > 0) ctr_atomic_add_global(a.sc_table + (int64_t)id * a.k + col, v);
          else if (id == 0) padacc[nb] += v;
        }
      }
      continue;
    }
    if constexpr (EPI == 1) {
      // mask by act'(xin) and per-group column sums.  The mask values of TWO column tiles are requested before
      // the first is used: every wait on an epilogue load also drains the ring's loads of the next tile, so one
      // dependent group of 16 loads per column tile stalled the pipeline four times per row tile (+500 us on
      // DIN's 3276800 x 64 x 128 layer).
      constexpr int NB = NT >= 2 ? 2 : 1;
      const int64_t first = i0 + 32 * wave;
      const int64_t g0 = a.gsum ? first / a.group : 0;
      const int64_t edge = (g0 + 1) * (int64_t)a.group;  // first row of the next group
#pragma unroll
      for (int nb0 = 0; nb0 < NT; nb0 += NB) {
        float xv[NB][16];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const int col = c0 + 32 * (nb0 + q) + r;
          if (a.xmask) {
            // register e holds rows (e & 3) + 8 (e >> 2) [+ 4 in the upper half-wave]: their words sit in those
            // lanes of mrow
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int rho = (e & 3) + 8 * (e >> 2);
              const uint32_t w0 = __builtin_amdgcn_readlane(mrow[nb0 + q], rho);
              const uint32_t w1 = __builtin_amdgcn_readlane(mrow[nb0 + q], rho + 4);
              xv[q][e] = (((h ? w1 : w0) >> r) & 1u) ? 1.0f : 0.0f;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int64_t i = first + (e & 3) + 8 * (e >> 2) + 4 * h;
              xv[q][e] = (a.xin && i < a.m && col < a.k) ? ctr_ldg(a.xin + i * a.ldxin + col) : 1.0f;
            }
          }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const int nb = nb0 + q;
          const int col = c0 + 32 * nb + r;
          if (col < a.k) {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              v[e] = acc[nb][0][e];
#pragma unroll
              for (int c = 1; c < CH; ++c) v[e] += acc[nb][c][e];
              if (a.xmask) v[e] *= xv[q][e];
              else if (a.xin) v[e] *= ctr_act_grad(xv[q][e], a.act_in);
            }
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int64_t i = first + (e & 3) + 8 * (e >> 2) + 4 * h;
              if (i < a.m) {
                ctr_stg(a.gx + i * a.ldgx + col, v[e]);
                if (i < edge) s0 += v[e];
                else s1 += v[e];
              }
            }
            if (a.gsum) {
              // the wave's 32 rows lie in at most two groups (group >= 32, checked on the host): fold the two
              // half-waves, one atomic per column and side
              s0 += __shfl_xor(s0, 32, 64);
              s1 += __shfl_xor(s1, 32, 64);
              if (h == 0 && first < a.m) {
                ctr_atomic_add_global(a.gsum + g0 * a.ldgsum + col, s0);
                if (edge < first + 32 && edge < a.m) ctr_atomic_add_global(a.gsum + (g0 + 1) * a.ldgsum + col, s1);
              }
            }
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
      const int col = c0 + 32 * nb + r;
      if (col < a.k) {
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          v[e] = acc[nb][0][e];
#pragma unroll
          for (int c = 1; c < CH; ++c) v[e] += acc[nb][c][e];
        }
        if (a.accumulate) {
          // all 16 old values in flight at once (the wait for them also drains the ring's loads)
          float old[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int64_t i = i0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * h;
            old[e] = i < a.m ? ctr_ldg(a.gx + i * a.ldgx + col) : 0.0f;
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] += old[e];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t i = i0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (i < a.m) ctr_stg(a.gx + i * a.ldgx + col, v[e]);
        }
      }
    }
  }
  if constexpr (EPI == 2) {
    // the workgroup's share of the padding row: half-waves fold by shuffle, waves through LDS (the ring is idle:
    // every wave has passed its last contraction step), one atomic per column
    __shared__ float s_pad[kThreads / 64][32 * NT];
    const int r = lane0 & 31, h = lane0 >> 5;
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
      float t = padacc[nb];
      t += __shfl_xor(t, 32, 64);
      if (h == 0) s_pad[wave][32 * nb + r] = t;
    }
    __syncthreads();
    for (int cidx = threadIdx.x; cidx < 32 * NT; cidx += kThreads) {
      float t = 0.0f;
      for (int wv = 0; wv < kThreads / 64; ++wv) t += s_pad[wv][cidx];
      if (t != 0.0f) ctr_atomic_add_global(a.sc_table + c0 + cidx, t);
    }
  }
}

}  // namespace

bool ctr_gemm_dlds_dx_ok(const float* w, int64_t ldw, const float* y, int64_t ldy, const float* gy, int64_t ldgy,
                         int64_t m, int n, int k, int act) {
  if (n < 4 || k < 4 || m < 1) return false;  // a chunk is fetched from n-4 / k-4 at the latest
  return act == CTR_ACT_NONE || y != nullptr;
}

struct DxEpilogue {
  bool on = false;
  const float* xin = nullptr; int64_t ldxin = 0; int act_in = CTR_ACT_NONE;
  float* gsum = nullptr; int64_t ldgsum = 0; int group = 1;
  const uint32_t* xmask = nullptr; int64_t ldxmask = 0;
  int mode = 1;   // 1: masked write-back + group sums, 2: scatter-add by row ids
  const int64_t* sc_idx = nullptr; const float* sc_attn = nullptr; const float* sc_gpool = nullptr; int64_t sc_ldgp = 0;
  float* sc_table = nullptr; int64_t sc_vocab = 0;
};

static int launch_dx(const float* w, int64_t ldw, const float* y, int64_t ldy, const float* gy, int64_t ldgy, float* gx,
                     int64_t ldgx, int accumulate, int64_t m, int n, int k, int act, hipStream_t st,
                     const DxEpilogue& ep = DxEpilogue()) {
  int nt = k <= 32 ? 1 : (k <= 64 ? 2 : 4);
  const int64_t mtiles = ctr_ceil_div(m, kBM);
  while (nt > 1 && !ep.on && mtiles * ctr_ceil_div(k, 32 * nt) < 128) nt >>= 1;   // few rows: see gemm_dlds.hip
  const int64_t ny = ctr_ceil_div(k, 32 * nt);
  // rounded down: a workgroup beyond the resident ones would start a second round.  Two per CU with a Y tile in the
  // ring (72 KB of LDS); without one (48 KB, <= 144 registers) three fit
  constexpr int wgs_plain = 3;
  int64_t gx_ = 256 * (act == CTR_ACT_NONE ? wgs_plain : 2) / ny;
  if (gx_ > mtiles) gx_ = mtiles;
  if (gx_ < 1) gx_ = 1;
  CTR_REQUIRE(ny <= 65535, CTR_ELIMIT);
  const DxArgs a{gy, ldgy, act == CTR_ACT_NONE ? nullptr : y, ldy, w, ldw, gx, ldgx, m, n, k, accumulate,
                 ep.xin, ep.ldxin, ep.act_in, ep.gsum, ep.ldgsum, ep.group, ep.xmask, ep.ldxmask,
                 ep.sc_idx, ep.sc_attn, ep.sc_gpool, ep.sc_ldgp, ep.sc_table, ep.sc_vocab};
  const dim3 grid((unsigned)gx_, (unsigned)ny);
#define CTR_DX(NT_, ACT_)                                                                                  \
  do {                                                                                                     \
    if (ep.on && ep.mode == 2) hipLaunchKernelGGL((gemm_dx_dlds_kernel<NT_, CTR_ACT_NONE, 2>), grid, dim3(kThreads), 0, st, a); \
    else if (ep.on) hipLaunchKernelGGL((gemm_dx_dlds_kernel<NT_, ACT_, 1>), grid, dim3(kThreads), 0, st, a); \
    else hipLaunchKernelGGL((gemm_dx_dlds_kernel<NT_, ACT_>), grid, dim3(kThreads), 0, st, a);             \
  } while (0)
#define CTR_DX_ACT(NT_)                                      \
  do {                                                       \
    if (act == CTR_ACT_NONE) CTR_DX(NT_, CTR_ACT_NONE);      \
    else if (act == CTR_ACT_RELU) CTR_DX(NT_, CTR_ACT_RELU); \
    else CTR_DX(NT_, CTR_ACT_SIGMOID);                       \
  } while (0)
  if (nt == 1) CTR_DX_ACT(1);
  else if (nt == 2) CTR_DX_ACT(2);
  else CTR_DX_ACT(4);
#undef CTR_DX_ACT
#undef CTR_DX
  return ctr_launch_status();
}

int ctr_gemm_dlds_dx(const float* w, int64_t ldw, const float* y, int64_t ldy, const float* gy, int64_t ldgy, float* gx,
                     int64_t ldgx, int accumulate, int64_t m, int n, int k, int act, hipStream_t st) {
  // a few input columns past a multiple of 128 get their own launch with a narrow tile (see gemm_dlds.hip);
  // the remainder needs >= 4 columns for its loads to stay inside W's rows, so it borrows from the main part
  int rem = k % 128;
  if (k > 128 && rem != 0 && rem <= 64) {
    if (rem < 4) rem += 32;
    const int main_k = k - rem;
    int rc = launch_dx(w, ldw, y, ldy, gy, ldgy, gx, ldgx, accumulate, m, n, main_k, act, st);
    if (rc != CTR_OK) return rc;
    return launch_dx(w + main_k, ldw, y, ldy, gy, ldgy, gx + main_k, ldgx, accumulate, m, n, rem, act, st);
  }
  return launch_dx(w, ldw, y, ldy, gy, ldgy, gx, ldgx, accumulate, m, n, k, act, st);
}

// C ABI (include/ctrhip.h): gX = ((gY * act'(Y)) W) * act_in'(Xin)  [or * sign bit of Xin];  gsum[row / group, :] += gX[row, :]
extern "C" int ctr_linear_dx_masked(const float* w, int64_t ldw, const float* y, int64_t ldy, const float* gy,
                                    int64_t ldgy, int act, const float* xin, int64_t ldxin, int act_in,
                                    const uint32_t* xmask /*nullable*/, int64_t ldxmask, float* gx, int64_t ldgx,
                                    float* gsum, int64_t ldgsum, int group, int64_t m, int n, int k, void* stream) {
  CTR_REQUIRE(m >= 0 && n >= 1 && k >= 1, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  CTR_REQUIRE(w && gy && gx && ldw >= k && ldgy >= n && ldgx >= k, CTR_EINVAL);
  CTR_REQUIRE(act >= CTR_ACT_NONE && act <= CTR_ACT_SIGMOID && act_in >= CTR_ACT_NONE && act_in <= CTR_ACT_SIGMOID, CTR_EINVAL);
  CTR_REQUIRE(act == CTR_ACT_NONE || (y && ldy >= n), CTR_EINVAL);
  CTR_REQUIRE(act_in == CTR_ACT_NONE || xmask || (xin && ldxin >= k), CTR_EINVAL);
  CTR_REQUIRE(!xmask || (act_in == CTR_ACT_RELU && k % 32 == 0 && ldxmask >= k / 32), CTR_EINVAL);  // bits = relu'(xin)
  CTR_REQUIRE(!gsum || (group >= 32 && ldgsum >= k), CTR_EINVAL);  // a wave's 32 rows span at most two groups
  CTR_REQUIRE(ctr_gemm_dlds_dx_ok(w, ldw, y, ldy, gy, ldgy, m, n, k, act) && k <= 128, CTR_ELIMIT);
  DxEpilogue ep;
  ep.on = true;
  ep.xin = (act_in == CTR_ACT_NONE || xmask) ? nullptr : xin;
  ep.xmask = xmask;
  ep.ldxmask = ldxmask;
  ep.ldxin = ldxin;
  ep.act_in = act_in;
  ep.gsum = gsum;
  ep.ldgsum = ldgsum;
  ep.group = gsum ? group : 1;
  return launch_dx(w, ldw, y, ldy, gy, ldgy, gx, ldgx, 0, m, n, k, act, (hipStream_t)stream, ep);
}

// C ABI (include/ctrhip.h): table[idx[i], :] += gY[i, :] W + attn[i] * gpool[i / group, :]   (gX itself is not stored)
extern "C" int ctr_linear_dx_scatter(const float* w, int64_t ldw, const float* gy, int64_t ldgy, const int64_t* idx,
                                     const float* attn, const float* gpool, int64_t ldgp, int group, float* table,
                                     int64_t vocab, int64_t m, int n, int k, void* stream) {
  CTR_REQUIRE(m >= 0 && n >= 1 && k >= 1, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  CTR_REQUIRE(w && gy && idx && attn && gpool && table && ldw >= k && ldgy >= n && ldgp >= k, CTR_EINVAL);
  CTR_REQUIRE(group >= 32 && vocab >= 1, CTR_EINVAL);  // a wave's 32 rows span at most two groups
  CTR_REQUIRE(k % 32 == 0 && k <= 128 && vocab < (1ll << 31) && m < (1ll << 40), CTR_ELIMIT);
  CTR_REQUIRE(ctr_gemm_dlds_dx_ok(w, ldw, nullptr, 0, gy, ldgy, m, n, k, CTR_ACT_NONE), CTR_ELIMIT);
  DxEpilogue ep;
  ep.on = true;
  ep.mode = 2;
  ep.group = group;
  ep.sc_idx = idx;
  ep.sc_attn = attn;
  ep.sc_gpool = gpool;
  ep.sc_ldgp = ldgp;
  ep.sc_table = table;
  ep.sc_vocab = vocab;
  return launch_dx(w, ldw, nullptr, 0, gy, ldgy, /*gx (unused)*/ table, k, 0, m, n, k, CTR_ACT_NONE, (hipStream_t)stream, ep);
}
