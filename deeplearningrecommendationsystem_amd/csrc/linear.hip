// Dense layers of the zoo on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact
// f32 products, f32 accumulate -- the parity bar is 1e-5 on logits, so no bf16).
//
// One templated tile kernel computes  Out[i][j] = sum_kk A(i,kk) * B(kk,j)  for a
// 128 x (32*NT) tile per 256-thread workgroup (4 waves, one 32-row strip each) with a
// 32-deep contraction step.  Each operand is staged global -> registers -> LDS in
// the layout its memory order gives for free:
//   KC ("contraction contiguous", e.g. X[m][k], W[n][k]):  LDS [row][kk], 36-float
//       row stride; a lane reads its 16 kk values with four conflict-free ds_read_b128;
//   KS ("contraction strided",   e.g. W[n][k] used as B(kk=n, j=k)): LDS [kk][row];
//       a lane reads one float per MFMA step, lanes on consecutive banks.
// The two lane halves of the 32x32x2 MFMA take kk = 16h + t (t = step), the same
// split for both operands, so only the order of the fp32 sum differs from k order.
// The next tile's global loads are issued before the MFMAs of the current one.
//   forward      Y  = act(X W^T + b (+R))   A = X  (KC)   B = W  (KC)
//   backward dX  gX = gZ W                  A = gZ (KC)   B = W  (KS)
//   backward dW  gW += gZ^T X               A = gZ (KS)   B = X  (KS), split over m, fp32 atomics
// with gZ = gY * act'(Y) formed while staging (never materialised).
#include "ctr_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kBM = 128;
constexpr int kBK = 16;             // contraction depth of one pipeline step
constexpr int kHalf = kBK / 2;      // values per lane and step: the lane halves take kk = kHalf*h + t
constexpr int kKcStride = kBK + 4;  // floats; 80 B rows: 8 consecutive rows cover all 32 banks with b128 reads
// resident workgroups per CU = waves per SIMD: four.  The forward (both operands KC) needs exactly
// 4 x 40960 B = the CU's 160 KB of LDS and fits the 128-register cap without a spill; dX / dW
// stage one or both operands [kk][row] (34-37 KB)
constexpr int wg_per_cu(int, int) { return 4; }

enum { KC = 0, KS = 1 };

// ---- sources: element (r, c) = p[r*ld + c]; zero outside [0,rows) x [0,cols) ----
struct PlainSrc {
  const float* p;
  int64_t ld;
  int64_t rows;
  int64_t cols;
  bool vec;  // base 16-B aligned and ld % 4 == 0
  __device__ __forceinline__ float4 load4(int64_t r, int64_t c) const {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) {
      const float* q = p + r * ld + c;
      if (vec && c + 3 < cols) {
        v = *reinterpret_cast<const float4*>(q);
      } else {
        if (c + 0 < cols) v.x = q[0];
        if (c + 1 < cols) v.y = q[1];
        if (c + 2 < cols) v.z = q[2];
        if (c + 3 < cols) v.w = q[3];
      }
    }
    return v;
  }
  // fast path: one predicated dwordx4 -- straight-line code the scheduler can slide under
  // the MFMAs.  Needs 16-B aligned rows; out-of-range rows / columns read as zero.  Only the
  // last partial vector of a row whose length is not a multiple of 4 takes the scalar tail.
  __device__ __forceinline__ float4 load4_fast(int64_t r, int64_t c) const {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) {
      const float* q = p + r * ld + c;
      if (c + 3 < cols) {
        v = *reinterpret_cast<const float4*>(q);
      } else if (c < cols) {
        v.x = q[0];
        if (c + 1 < cols) v.y = q[1];
        if (c + 2 < cols) v.z = q[2];
      }
    }
    return v;
  }
  __device__ __forceinline__ bool fast_ok() const { return vec; }
};

// gZ = gY * act'(Y), formed on the fly from the saved layer output
struct GzSrc {
  PlainSrc gy;
  PlainSrc y;
  int act;
  __device__ __forceinline__ float4 load4(int64_t r, int64_t c) const {
    float4 g = gy.load4(r, c);
    if (act != CTR_ACT_NONE) {
      const float4 o = y.load4(r, c);
      g.x *= ctr_act_grad(o.x, act);
      g.y *= ctr_act_grad(o.y, act);
      g.z *= ctr_act_grad(o.z, act);
      g.w *= ctr_act_grad(o.w, act);
    }
    return g;
  }
  __device__ __forceinline__ bool fast_ok() const { return gy.fast_ok() && (act == CTR_ACT_NONE || y.fast_ok()); }
  __device__ __forceinline__ float4 load4_fast(int64_t r, int64_t c) const {
    float4 g = gy.load4_fast(r, c);
    if (act != CTR_ACT_NONE) {
      const float4 o = y.load4_fast(r, c);
      g.x *= ctr_act_grad(o.x, act);
      g.y *= ctr_act_grad(o.y, act);
      g.z *= ctr_act_grad(o.z, act);
      g.w *= ctr_act_grad(o.w, act);
    }
    return g;
  }
};

// ---- epilogues: consume one accumulator element at (i, j) ----
struct FwdEpi {
  float* y;
  int64_t ldy;
  const float* bias;
  const float* res;
  int64_t ldr;
  int act;
  __device__ __forceinline__ void operator()(int64_t i, int64_t j, float acc) const {
    float z = acc;
    if (bias) z += bias[j];
    if (res) z += res[i * ldr + j];
    y[i * ldy + j] = ctr_act(z, act);
  }
};
struct StoreEpi {
  float* p;
  int64_t ld;
  int accumulate;
  __device__ __forceinline__ void operator()(int64_t i, int64_t j, float acc) const {
    float* q = p + i * ld + j;
    *q = accumulate ? *q + acc : acc;
  }
};
// dW pass with many row chunks: chunk z stores its partial tile into slab z of the
// workspace with plain stores; reduce.hip sums the slabs (fixed order, no atomics)
struct SlabEpi {
  float* ws;
  int64_t ld;
  int64_t slab_stride;
  __device__ __forceinline__ void operator()(int64_t i, int64_t j, float acc) const {
    ws[(int64_t)blockIdx.z * slab_stride + i * ld + j] = acc;
  }
};
// the same for the swapped dW product (tile rows = input features, tile columns = output units):
// element (i, j) of the tile is gW[j][i]
struct SlabEpiT {
  float* ws;
  int64_t ld;  // = k, the row length of gW
  int64_t slab_stride;
  __device__ __forceinline__ void operator()(int64_t i, int64_t j, float acc) const {
    ws[(int64_t)blockIdx.z * slab_stride + j * ld + i] = acc;
  }
};
struct AtomicEpi {
  float* p;
  int64_t ld;
  __device__ __forceinline__ void operator()(int64_t i, int64_t j, float acc) const {
    unsafeAtomicAdd(p + i * ld + j, acc);
  }
};

// Stage a ROWS x 32 (KC) or 32 x ROWS (KS) tile.  `row0` is the first output row /
// column of the tile, `k0` the first contraction index.
template <int MODE, int ROWS, class Src>
struct Stager {
  static constexpr int kUnits = ROWS * kBK / 4;                           // float4 in the tile
  static constexpr int kVecs = (kUnits + kThreads - 1) / kThreads;        // float4 per thread
  static constexpr bool kPartial = kUnits % kThreads != 0;                // narrow tiles: some threads idle
  static constexpr int kKsStride = ROWS + 4;
  static constexpr int kLdsFloats = MODE == KC ? ROWS * kKcStride : kBK * kKsStride;
  float4 reg[kVecs];

  // `fast`: workgroup-uniform, true when the whole tile is inside the matrix and 16-B aligned
  __device__ __forceinline__ void load(const Src& s, int64_t row0, int64_t k0, bool fast) {
    const int t = threadIdx.x;
    if (fast) {
#pragma unroll
      for (int p = 0; p < kVecs; ++p) {
        if (kPartial && t + kThreads * p >= kUnits) {
          reg[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (MODE == KC) {
          const int c4 = t % (kBK / 4), r = t / (kBK / 4) + (kThreads / (kBK / 4)) * p;
          reg[p] = s.load4_fast(row0 + r, k0 + c4 * 4);
        } else {
          constexpr int kPerRow = ROWS / 4;  // float4 per kk row
          const int v = t + kThreads * p;
          const int c4 = v % kPerRow, kk = v / kPerRow;
          reg[p] = s.load4_fast(k0 + kk, row0 + c4 * 4);
        }
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < kVecs; ++p) {
      if (kPartial && t + kThreads * p >= kUnits) {
        reg[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (MODE == KC) {
        const int c4 = t % (kBK / 4), r = t / (kBK / 4) + (kThreads / (kBK / 4)) * p;
        reg[p] = s.load4(row0 + r, k0 + c4 * 4);
      } else {
        constexpr int kPerRow = ROWS / 4;
        const int v = t + kThreads * p;
        const int c4 = v % kPerRow, kk = v / kPerRow;
        reg[p] = s.load4(k0 + kk, row0 + c4 * 4);
      }
    }
  }
  __device__ __forceinline__ void store(float* lds) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < kVecs; ++p) {
      if (kPartial && t + kThreads * p >= kUnits) {
        // idle thread of a narrow tile
      } else if (MODE == KC) {
        const int c4 = t % (kBK / 4), r = t / (kBK / 4) + (kThreads / (kBK / 4)) * p;
        *reinterpret_cast<float4*>(lds + r * kKcStride + c4 * 4) = reg[p];
      } else {
        constexpr int kPerRow = ROWS / 4;
        const int v = t + kThreads * p;
        const int c4 = v % kPerRow, kk = v / kPerRow;
        *reinterpret_cast<float4*>(lds + kk * kKsStride + c4 * 4) = reg[p];
      }
    }
  }
};

// the kHalf contraction values (kk = kHalf*h .. kHalf*h + kHalf-1) of tile row `r` for this lane
template <int MODE, int ROWS>
__device__ __forceinline__ void read_frag(const float* lds, int r, int h, float (&f)[kHalf]) {
  if (MODE == KC) {
    const float4* q = reinterpret_cast<const float4*>(lds + r * kKcStride + kHalf * h);
#pragma unroll
    for (int v = 0; v < kHalf / 4; ++v) {
      const float4 x = q[v];
      f[4 * v + 0] = x.x; f[4 * v + 1] = x.y; f[4 * v + 2] = x.z; f[4 * v + 3] = x.w;
    }
  } else {
    constexpr int kKsStride = ROWS + 4;
#pragma unroll
    for (int t = 0; t < kHalf; ++t) f[t] = lds[(kHalf * h + t) * kKsStride + r];
  }
}

template <int NT, int AMODE, int BMODE, class ASrc, class BSrc, class Epi>
__global__ void __launch_bounds__(kThreads, wg_per_cu(AMODE, BMODE))
gemm_tile_kernel(const ASrc a, const BSrc b, const Epi epi, int64_t M, int N, int64_t K, int64_t k_chunk,
                 float* __restrict__ bias_grad /* dW pass only: column sums of A (or of B, see below) */,
                 int64_t bias_slab_stride /* 0: atomics into bias_grad, else slab z of it */,
                 int bias_from_b /* swapped dW: gZ is the B operand, its column sums come from the B tile */) {
  constexpr int BN = 32 * NT;
  using AStage = Stager<AMODE, kBM, ASrc>;
  using BStage = Stager<BMODE, BN, BSrc>;
  // two LDS buffers per operand: tile k+1 is written while tile k is multiplied, one
  // barrier per 32-deep step
  __shared__ __attribute__((aligned(16))) float s_a[2][AStage::kLdsFloats];
  __shared__ __attribute__((aligned(16))) float s_b[2][BStage::kLdsFloats];

  // Persistent over the M tiles: workgroup x walks tiles x, x + gridDim.x, ... and the
  // global-load / LDS-store / MFMA pipeline runs straight across tile boundaries, so only
  // the very first tile pays an exposed load latency and every epilogue overlaps the loads
  // of the next tile.  With short contractions (k = 128: four steps per tile) the per-tile
  // prologue used to cost more than the MFMAs.
  const int64_t mtiles = (M + kBM - 1) / kBM;
  const int64_t j0 = (int64_t)blockIdx.y * BN;
  const int64_t kb = (int64_t)blockIdx.z * k_chunk;
  const int64_t ke = kb + k_chunk < K ? kb + k_chunk : K;
  const int nk = (int)((ke - kb + kBK - 1) / kBK);

  const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  // Four independent accumulator chains per wave whatever NT is: consecutive MFMAs never
  // wait on each other's result.  NT = 4: one chain per column tile; NT = 2 / 1: the
  // contraction steps are dealt round-robin to 2 / 4 chains per tile, summed at the end.
  constexpr int CH = 4 / NT;
  floatx16 acc[NT][CH];

  // aligned operands take the branch-free predicated load path for every tile
  const bool a_in = a.fast_ok();
  const bool b_in = b.fast_ok();

  AStage sa;
  BStage sb;
  int64_t tile = blockIdx.x;
  if (tile >= mtiles || nk <= 0) return;
  // (tile, step) of the pipeline slot one / two steps ahead of the one being multiplied
  auto advance = [&](int64_t& t, int& k) {
    if (++k == nk) {
      k = 0;
      t += gridDim.x;
    }
  };
  int64_t t1 = tile, t2 = tile;
  int k1 = 0, k2 = 0;
  advance(t1, k1);
  t2 = t1;
  k2 = k1;
  advance(t2, k2);
  sa.load(a, tile * kBM, kb, a_in);
  sb.load(b, j0, kb, b_in);
  sa.store(s_a[0]);
  sb.store(s_b[0]);
  if (t1 < mtiles) {
    sa.load(a, t1 * kBM, kb + (int64_t)k1 * kBK, a_in);
    sb.load(b, j0, kb + (int64_t)k1 * kBK, b_in);
  }
  __syncthreads();
  int cur = 0;
  for (; tile < mtiles; tile += gridDim.x) {
    const int64_t i0 = tile * kBM;
    // lane-derived offsets (64 output addresses per lane in the epilogue) are invariant across
    // tiles; hoisted out of this loop they cost ~120 registers and the second wave per SIMD
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][c][e] = 0.0f;
    float colsum = 0.0f;
    for (int ks = 0; ks < nk; ++ks) {
      if (t1 < mtiles) {
        sa.store(s_a[cur ^ 1]);
        sb.store(s_b[cur ^ 1]);
      }
      if (t2 < mtiles) {
        sa.load(a, t2 * kBM, kb + (int64_t)k2 * kBK, a_in);
        sb.load(b, j0, kb + (int64_t)k2 * kBK, b_in);
      }
      advance(t1, k1);
      advance(t2, k2);
      float fa[kHalf];
      read_frag<AMODE, kBM>(s_a[cur], 32 * wave + r, h, fa);
      float fb[NT][kHalf];
#pragma unroll
      for (int n = 0; n < NT; ++n) read_frag<BMODE, BN>(s_b[cur], 32 * n + r, h, fb[n]);
#pragma unroll
      for (int t = 0; t < kHalf; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[n][t % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[n][t], acc[n][t % CH], 0, 0, 0);
      if (bias_grad != nullptr && AMODE == KS && BMODE == KS) {
        // the tile of gZ is [kk][row]: thread `row` adds the step's contraction values
        if (!bias_from_b && blockIdx.y == 0 && threadIdx.x < kBM) {
          constexpr int kKsStride = kBM + 4;
#pragma unroll 8
          for (int kk = 0; kk < kBK; ++kk) colsum += s_a[cur][kk * kKsStride + threadIdx.x];
        }
        if (bias_from_b && tile == 0 && threadIdx.x < BN) {
          constexpr int kKsStrideB = BN + 4;
#pragma unroll 8
          for (int kk = 0; kk < kBK; ++kk) colsum += s_b[cur][kk * kKsStrideB + threadIdx.x];
        }
      }
      __syncthreads();
      cur ^= 1;
    }

    // C/D map of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int64_t j = j0 + 32 * n + r;
      if (j < N) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t i = i0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * h;
          float v = acc[n][0][e];
#pragma unroll
          for (int c = 1; c < CH; ++c) v += acc[n][c][e];
          if (i < M) epi(i, j, v);
        }
      }
    }
    if (bias_grad != nullptr && AMODE == KS && BMODE == KS) {
      const bool mine = bias_from_b ? (tile == 0 && threadIdx.x < BN) : (blockIdx.y == 0 && threadIdx.x < kBM);
      const int64_t i = (bias_from_b ? j0 : i0) + threadIdx.x;
      if (mine && i < (bias_from_b ? (int64_t)N : M)) {
        if (bias_slab_stride)
          bias_grad[(int64_t)blockIdx.z * bias_slab_stride + i] = colsum;
        else
          unsafeAtomicAdd(bias_grad + i, colsum);
      }
    }
  }
}

inline PlainSrc plain(const float* p, int64_t ld, int64_t rows, int64_t cols) {
  PlainSrc s;
  s.p = p;
  s.ld = ld;
  s.rows = rows;
  s.cols = cols;
  s.vec = ctr_aligned16(p) && (ld % 4 == 0);
  return s;
}

inline int pick_nt(int n) { return n <= 32 ? 1 : (n <= 64 ? 2 : 4); }

// contraction chunks the launch really makes (chunks are whole 32-deep steps)
inline int effective_splits(int64_t K, int64_t splits) {
  const int64_t k_chunk = ctr_ceil_div(ctr_ceil_div(K, splits), kBK) * kBK;
  return (int)ctr_ceil_div(K, k_chunk);
}

template <int AMODE, int BMODE, class ASrc, class BSrc, class Epi>
int launch(const ASrc& a, const BSrc& b, const Epi& e, int64_t M, int N, int64_t K, int splits, float* bias_grad,
           hipStream_t st, int64_t bias_slab_stride = 0, int bias_from_b = 0) {
  int nt = pick_nt(N);
  const int64_t k_chunk = ctr_ceil_div(ctr_ceil_div(K, splits), kBK) * kBK;
  const int zs = (int)ctr_ceil_div(K, k_chunk);
  // as many workgroups as stay resident, each walking its share of the M tiles
  const int64_t mtiles = ctr_ceil_div(M, kBM);
  // a handful of tiles for 256 CUs (products over ~1000 table rows): narrower column tiles instead
  while (nt > 1 && mtiles * ctr_ceil_div(N, 32 * nt) * zs < 128) nt >>= 1;
  const int64_t others = ctr_ceil_div(N, 32 * nt) * zs;
  int64_t gx = ctr_ceil_div(256 * wg_per_cu(AMODE, BMODE), others);
  if (gx > mtiles) gx = mtiles;
  if (gx < 1) gx = 1;
  const dim3 grid((unsigned)gx, (unsigned)ctr_ceil_div(N, 32 * nt), (unsigned)zs);
  CTR_REQUIRE(grid.y <= 65535 && grid.z <= 65535, CTR_ELIMIT);
  switch (nt) {
    case 1:
      hipLaunchKernelGGL((gemm_tile_kernel<1, AMODE, BMODE, ASrc, BSrc, Epi>), grid, dim3(kThreads), 0, st, a, b, e, M,
                         N, K, k_chunk, bias_grad, bias_slab_stride, bias_from_b);
      break;
    case 2:
      hipLaunchKernelGGL((gemm_tile_kernel<2, AMODE, BMODE, ASrc, BSrc, Epi>), grid, dim3(kThreads), 0, st, a, b, e, M,
                         N, K, k_chunk, bias_grad, bias_slab_stride, bias_from_b);
      break;
    default:
      hipLaunchKernelGGL((gemm_tile_kernel<4, AMODE, BMODE, ASrc, BSrc, Epi>), grid, dim3(kThreads), 0, st, a, b, e, M,
                         N, K, k_chunk, bias_grad, bias_slab_stride, bias_from_b);
      break;
  }
  return ctr_launch_status();
}

}  // namespace

// Output widths a little above a multiple of the 128-column tile (641 = 5 * 128 + 1: Deep & Cross at emb 128): a sixth of
// the tiles would compute padding.  The last 1..16 columns go to a launch of their own with a 32-column tile (it streams
// the rows again: a long batch only).  Measured (65536 x 641 x 641, profiles/r03_tail_split.txt): forward 652 -> 578 us.
// The same split of the INPUT-gradient's columns measured no gain (the 32-column tile kernel streams gZ at ~1.4 TB/s:
// 610 -> 629 us) and a 33-column tail (161 = 128 + 33, Deep Crossing) on a 64-column tile neither: not done.
static inline int tail_cols(int64_t m, int cols) {
  const int r = cols % 128;
  return (m >= 4096 && cols > 128 && r >= 1 && r <= 16) ? r : 0;
}

extern "C" int ctr_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                              const float* residual, int64_t ldr, float* y, int64_t ldy, int64_t m, int n, int k,
                              int act, void* stream) {
  CTR_REQUIRE(m >= 0 && n > 0 && k > 0, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  CTR_REQUIRE(x && w && y, CTR_EINVAL);
  CTR_REQUIRE(ldx >= k && ldw >= k && ldy >= n && (!residual || ldr >= n), CTR_EINVAL);
  CTR_REQUIRE(act >= CTR_ACT_NONE && act <= CTR_ACT_SIGMOID, CTR_EINVAL);
  if (const int r = tail_cols(m, n)) {
    const int n0 = n - r;
    int rc = ctr_linear_fwd(x, ldx, w, ldw, bias, residual, ldr, y, ldy, m, n0, k, act, stream);
    if (rc != CTR_OK) return rc;
    return ctr_linear_fwd(x, ldx, w + (int64_t)n0 * ldw, ldw, bias ? bias + n0 : nullptr, residual ? residual + n0 : nullptr,
                          ldr, y + n0, ldy, m, r, k, act, stream);
  }
  if (n == 1 && ctr_n1_supported(k))
    return ctr_n1_fwd(x, ldx, w, bias, residual, ldr, y, ldy, m, k, act, (hipStream_t)stream);
  // wide layer, long batch: the 256 x 256 macro tile (gemm_wide.hip)
  if (!residual && ctr_gemm_wide_ok(x, ldx, w, ldw, m, n, k))
    return ctr_gemm_wide_fwd(x, ldx, w, ldw, bias, y, ldy, m, n, k, act, (hipStream_t)stream);
  // K >= 16: operands stream global -> LDS directly (gemm_dlds.hip: 1.1-1.6x the tile kernel)
  if (ctr_gemm_dlds_ok(x, ldx, w, ldw, m, n, k))
    return ctr_gemm_dlds_fwd(x, ldx, w, ldw, bias, residual, ldr, y, ldy, m, n, k, act, (hipStream_t)stream);
  FwdEpi e{y, ldy, bias, residual, ldr, act};
  return launch<KC, KC>(plain(x, ldx, m, k), plain(w, ldw, n, k), e, m, n, k, 1, nullptr, (hipStream_t)stream);
}

extern "C" int ctr_linear_bwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* y, int64_t ldy,
                              const float* gy, int64_t ldgy, float* gx, int64_t ldgx, int accumulate_gx, float* gw,
                              int64_t ldgw, float* gb, int64_t m, int n, int k, int act, float* workspace,
                              int64_t workspace_floats, void* stream) {
  CTR_REQUIRE(m >= 0 && n > 0 && k > 0, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  CTR_REQUIRE(gy && ldgy >= n, CTR_EINVAL);
  CTR_REQUIRE(act >= CTR_ACT_NONE && act <= CTR_ACT_SIGMOID, CTR_EINVAL);
  CTR_REQUIRE(act == CTR_ACT_NONE || (y && ldy >= n), CTR_EINVAL);
  CTR_REQUIRE(!gx || (w && ldw >= k && ldgx >= k), CTR_EINVAL);
  CTR_REQUIRE(!(gw || gb) || (x && ldx >= k), CTR_EINVAL);
  CTR_REQUIRE(!gw || ldgw >= k, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  hipStream_t st = (hipStream_t)stream;
  if (n == 1 && ctr_n1_supported(k) && (gx || gw || gb))
    return ctr_n1_bwd(x, ldx, w, y, ldy, gy, ldgy, gx, ldgx, accumulate_gx, gw, gb, m, k, act, workspace,
                      workspace_floats, st);
  // few units on both sides over a long batch: dW streams (linear_skinny.hip); forward and dX stay on
  // the tile kernel, which beat the streaming variants tried for them
  if (gw && workspace && ctr_skinny_dw_ok(x, ldx, y, ldy, gy, ldgy, gw, ldgw, m, n, k, act)) {
    int rc = ctr_skinny_dw(x, ldx, y, ldy, gy, ldgy, gw, gb, m, n, k, act, workspace, workspace_floats, st);
    if (rc != CTR_OK || !gx) return rc;
    gw = nullptr;
    gb = nullptr;
  }
  // wide layers with aligned operands: gY, Y and X stream global -> LDS directly (gemm_dlds_dw.hip)
  if (gw && workspace && ctr_gemm_dlds_dw_ok(x, ldx, y, ldy, gy, ldgy, gw, ldgw, m, n, k, act)) {
    int rc = ctr_gemm_dlds_dw(x, ldx, y, ldy, gy, ldgy, gw, gb, m, n, k, act, workspace, workspace_floats, st);
    if (rc == CTR_OK) {
      if (!gx) return rc;
      gw = nullptr;
      gb = nullptr;
    } else if (rc != CTR_ELIMIT) {
      return rc;
    }
  }
  GzSrc gz;
  gz.gy = plain(gy, ldgy, m, n);
  gz.y = plain(y ? y : gy, y ? ldy : ldgy, m, n);
  gz.act = act;
  if (gx && ctr_gemm_dlds_dx_ok(w, ldw, y, ldy, gy, ldgy, m, n, k, act)) {
    // aligned operands, n % 16 == 0: gY, Y and W stream global -> LDS directly (gemm_dlds_dx.hip)
    int rc = ctr_gemm_dlds_dx(w, ldw, y, ldy, gy, ldgy, gx, ldgx, accumulate_gx, m, n, k, act, st);
    if (rc != CTR_OK) return rc;
  } else if (gx) {
    // gX[m][kcol] = sum_n gZ[m][n] W[n][kcol]: contraction = n
    StoreEpi e{gx, ldgx, accumulate_gx};
    int rc = launch<KC, KS>(gz, plain(w, ldw, n, k), e, m, k, n, 1, nullptr, st);
    if (rc != CTR_OK) return rc;
  }
  if (gw || gb) {
    // gW[n][kcol] += sum_m gZ[m][n] X[m][kcol]: contraction = m, split over workgroups
    GzSrc gzt = gz;  // indexed (r = m, c = n): KS operand of the transposed product
    // (tile count of whichever orientation the slab path below will pick)
    const int64_t tiles_nk = ctr_ceil_div(n, kBM) * ctr_ceil_div(k, 32 * pick_nt(k));
    const int64_t tiles_kn = ctr_ceil_div(k, kBM) * ctr_ceil_div(n, 32 * pick_nt(n));
    const int64_t tiles = tiles_kn < tiles_nk ? tiles_kn : tiles_nk;
    // enough row chunks for ~6 resident workgroups per CU: the chunks stream gY, Y and X once
    // and only occupancy hides their HBM latency (256 chunks = 1 workgroup per CU ran 4x slower)
    const int64_t target = tiles >= 4 ? 1024 : 1536;
    int64_t splits = (target + tiles - 1) / tiles;
    const int64_t max_splits = ctr_ceil_div(m, 128);   // at least 128 rows each
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    CTR_REQUIRE(gw != nullptr, CTR_EINVAL);  // gb without gw is not used by any model
    // every chunk adds its partial to the same n*k outputs.  More than a handful of
    // chunks -> per-chunk slabs in the workspace + one reduction pass; same-address
    // atomic chains are serialised by the memory side (~60 ns per link)
    const int64_t slab = (int64_t)n * k;
    if (workspace && splits > 8) {
      // as many row chunks as the workspace has slabs for
      const int64_t fit = workspace_floats / (slab + n);
      if (fit >= 8 && splits > fit) splits = fit;
    }
    splits = effective_splits(m, splits);
    const int64_t need = splits * (slab + n);
    // (a strided gW of more rows than the reduction pass has segments for: the atomic epilogue below takes any ldgw)
    const bool seg_ok = ldgw == k || n + 1 <= CTR_MAX_SEGMENTS;
    if (splits > 8 && workspace && workspace_floats >= need && seg_ok) {
      float* ws_b = workspace + splits * slab;
      // orientation: the 128-row side of the tile should be the longer of (n, k).  A funnel layer
      // (n = k/2, n < 128) fills half of a 128 x k tile as gZ^T X but all of a k x n tile as X^T gZ.
      auto padded = [](int64_t rows, int64_t cols) {
        const int nt = pick_nt((int)cols);
        return ctr_ceil_div(rows, kBM) * kBM * ctr_ceil_div(cols, 32 * nt) * 32 * nt;
      };
      int rc;
      if (padded(k, n) < padded(n, k)) {
        SlabEpiT et{workspace, k, slab};
        rc = launch<KS, KS>(plain(x, ldx, m, k), gzt, et, k, n, m, (int)splits, gb ? ws_b : nullptr, st, n, 1);
      } else {
        SlabEpi e{workspace, k, slab};
        rc = launch<KS, KS>(gzt, plain(x, ldx, m, k), e, n, k, m, (int)splits, gb ? ws_b : nullptr, st, n);
      }
      if (rc != CTR_OK) return rc;
      CtrSegments segs;
      segs.n = 0;
      if (ldgw == k) {
        segs.s[segs.n++] = CtrSegment{0, slab, gw};
      } else {
        for (int r = 0; r < n; ++r) segs.s[segs.n++] = CtrSegment{(int64_t)r * k, k, gw + r * ldgw};
      }
      rc = ctr_reduce_segments(workspace, (int)splits, slab, segs, st);
      if (rc != CTR_OK || !gb) return rc;
      CtrSegments bsegs;
      bsegs.n = 1;
      bsegs.s[0] = CtrSegment{0, n, gb};
      return ctr_reduce_segments(ws_b, (int)splits, n, bsegs, st);
    }
    if (splits > 64) splits = 64;  // no workspace: keep the same-address atomic chains short
    AtomicEpi e{gw, ldgw};
    int rc = launch<KS, KS>(gzt, plain(x, ldx, m, k), e, n, k, m, (int)splits, gb, st);
    if (rc != CTR_OK) return rc;
  }
  return CTR_OK;
}
