// Weight gradient  gW[n, :k] += sum_m gZ[m, n] X[m, :k],  gb[n] += sum_m gZ[m, n],  gZ = gY * act'(Y),
// with the three streamed operands (gY, Y, X) copied global -> LDS directly (global_load_lds_dwordx4)
// through a three-stage ring, like the forward kernel of gemm_dlds.hip.
//
// The contraction runs over the batch, so both MFMA operands are read "down the rows" of their
// row-major tiles: lane (r, h) takes element [8h + t][32*tile + r] for t = 0..7 -- 32 consecutive
// floats per half-wave and instruction, conflict-free without any swizzle, and the LDS image is the
// plain row-major tile the direct loads produce.  act'(Y) is applied to the gY values as they are
// read; rows past the end of a workgroup's row range are zeroed there too (their loads are clamped
// to a valid row).  The 16-byte loads only need 4-byte aligned addresses; see fetch() for the column tails.
//
// Output tile: 128 x 32*NT per 256-thread workgroup (which side is units: see SWAP), the batch cut into
// `parts` row ranges (multiples of 16 rows); every workgroup stores its partial tile into its part's
// slab of the workspace and reduce.hip sums the parts in a fixed order.
//
// The direct loads are issued from inline asm, which keeps them out of the compiler's waitcnt
// bookkeeping (it would otherwise put vmcnt(0) in front of every LDS read); the waits are explicit.
#include "ctr_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kAW = 128;  // units per workgroup
constexpr int kBK = 16;   // batch rows per pipeline step
constexpr int kStages = 3;

struct DwArgs {
  const float* gy; int64_t ldgy;
  const float* y; int64_t ldy;  // null: no activation
  const float* x; int64_t ldx;
  int64_t m; int n; int k; int act;
  int64_t rows_per_part;
  float* ws; int64_t slab;  // slab = n*k (+ n when bias)
  int want_bias;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}

// one wave instruction: lane L's 16 bytes land at lds_base + 16 L
// (m0 is a reserved register: the compiler only sets it right in front of an instruction that reads it,
// never keeps a value there, and rejects it as a clobber)
__device__ __forceinline__ void dma16(const float* g, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_base) : "memory");
}

__device__ __forceinline__ uint32_t lds_addr(const float* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

// copy rows [row0, row0+16) x columns [col0, col0+W) of a row-major matrix into a [16][W] stage
template <int W>
__device__ __forceinline__ void fetch(float* stage, const float* __restrict__ src, int64_t ld, int64_t row0,
                                      int64_t row_last, int col0, int cols_total, int lane, int wave) {
  constexpr int kPerRow = W / 4, kChunks = 16 * kPerRow;
  constexpr int kIters = (kChunks + kThreads - 1) / kThreads;
#pragma unroll
  for (int i = 0; i < kIters; ++i) {
    int q0 = 64 * wave + kThreads * i;
    if (kChunks % kThreads != 0 && q0 >= kChunks) q0 -= kChunks;  // narrow tile: fetched twice, same bytes
    const int q = q0 + lane;
    const int row = q / kPerRow, cc = q % kPerRow;
    int64_t gr = row0 + row;
    gr = gr < row_last ? gr : row_last;
    // a chunk that would cross the end of the row (or lies past it) is fetched from 4 floats before
    // the end: no load leaves the matrix, the reader adds the shift (tail_shift), the rest is dropped
    int col = col0 + cc * 4;
    col = col < cols_total - 4 ? col : cols_total - 4;
    dma16(src + gr * ld + col, __builtin_amdgcn_readfirstlane(lds_addr(stage + q0 * 4)));
  }
}

// SWAP = false: the 128-wide side of the tile are units (operand gZ, columns n0..), the 32*NT side inputs (X).
// SWAP = true : the 128-wide side are inputs (X, columns k0..), the 32*NT side units (gZ) -- for layers with
//               fewer than 128 units, which would leave half of the unswapped tile empty.
// logical column `col` of a matrix with `total` columns sits this many floats further right in its
// (shifted) last chunk
__device__ __forceinline__ int tail_shift(int col, int total) {
  return (total & 3) && col >= (total & ~3) && col < total ? 4 - (total & 3) : 0;
}

template <int NT, int ACT, bool SWAP>
__global__ void __launch_bounds__(kThreads, 2)
gemm_dw_dlds_kernel(const DwArgs a) {
  constexpr int BW = 32 * NT;
  constexpr int ZW = SWAP ? BW : kAW;  // width of the gY / Y tiles
  constexpr int XW = SWAP ? kAW : BW;  // width of the X tile
  constexpr bool has_y = ACT != CTR_ACT_NONE;
  __shared__ __attribute__((aligned(16))) float s_gy[kStages][kBK * ZW];
  __shared__ __attribute__((aligned(16))) float s_y[has_y ? kStages : 1][has_y ? kBK * ZW : 4];
  __shared__ __attribute__((aligned(16))) float s_x[kStages][kBK * XW];
  constexpr int kLoadsZ = (16 * (ZW / 4) + kThreads - 1) / kThreads;
  constexpr int kLoadsX = (16 * (XW / 4) + kThreads - 1) / kThreads;
  constexpr int kPerStep = (has_y ? 2 : 1) * kLoadsZ + kLoadsX;

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  // blockIdx.y walks the 128-wide side, blockIdx.z the 32*NT side
  const int n0 = SWAP ? blockIdx.z * BW : blockIdx.y * kAW;
  const int k0 = SWAP ? blockIdx.y * kAW : blockIdx.z * BW;
  const int64_t mb = (int64_t)blockIdx.x * a.rows_per_part;
  const int64_t me = mb + a.rows_per_part < a.m ? mb + a.rows_per_part : a.m;
  const int steps = (int)((me - mb + kBK - 1) / kBK);

  // LDS columns of this lane's operand elements (A side: 32*wave + r, B side: 32*nb + r)
  const int acol = 32 * wave + r + tail_shift((SWAP ? k0 : n0) + 32 * wave + r, SWAP ? a.k : a.n);
  int bcol[NT];
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) bcol[nb] = 32 * nb + r + tail_shift((SWAP ? n0 : k0) + 32 * nb + r, SWAP ? a.n : a.k);

  floatx16 acc[NT];
#pragma unroll
  for (int nb = 0; nb < NT; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nb][e] = 0.0f;
  float bsum[SWAP ? NT : 1];
#pragma unroll
  for (int i = 0; i < (SWAP ? NT : 1); ++i) bsum[i] = 0.0f;

  auto issue = [&](int stage, int s) {
    const int64_t row0 = mb + (int64_t)s * kBK;
    fetch<ZW>(s_gy[stage], a.gy, a.ldgy, row0, a.m - 1, n0, a.n, lane, wave);
    if (has_y) fetch<ZW>(s_y[stage], a.y, a.ldy, row0, a.m - 1, n0, a.n, lane, wave);
    fetch<XW>(s_x[stage], a.x, a.ldx, row0, a.m - 1, k0, a.k, lane, wave);
  };
  if (steps > 0) issue(0, 0);
  if (steps > 1) issue(1, 1);
  int stage = 0;
  for (int s = 0; s < steps; ++s) {
    // this wave's loads of step s have landed (those of step s+1 may still fly), then everybody's
    // have, and everybody has finished reading the stage refilled below
    if (s + 1 < steps) wait_vmcnt<kPerStep>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int refill = stage + 2;
    refill = refill >= kStages ? refill - kStages : refill;
    if (s + 2 < steps) issue(refill, s + 2);

    const int64_t row = mb + (int64_t)s * kBK + 8 * h;
    // gZ element [8h + t][col] of the stage, zero past the end of the row range
    auto gz = [&](int t, int col) {
      float g = s_gy[stage][(8 * h + t) * ZW + col];
      if (has_y) g *= ctr_act_grad(s_y[stage][(8 * h + t) * ZW + col], ACT);
      return row + t < me ? g : 0.0f;
    };
    float fa[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (SWAP) {
        fa[t] = s_x[stage][(8 * h + t) * XW + acol];
      } else {
        fa[t] = gz(t, acol);
        bsum[0] += fa[t];
      }
    }
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
      float fb[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (SWAP) {
          fb[t] = gz(t, bcol[nb]);
          bsum[nb] += fb[t];
        } else {
          fb[t] = s_x[stage][(8 * h + t) * XW + bcol[nb]];
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc[nb], 0, 0, 0);
    }
    stage = stage + 1 == kStages ? 0 : stage + 1;
  }
  // C/D map: column (B side) = lane & 31, row (A side) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* out = a.ws + (int64_t)blockIdx.x * a.slab;
#pragma unroll
  for (int nb = 0; nb < NT; ++nb) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int arow = 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * h, bcol = 32 * nb + r;
      const int un = SWAP ? n0 + bcol : n0 + arow;
      const int kc = SWAP ? k0 + arow : k0 + bcol;
      if (un < a.n && kc < a.k) ctr_stg(out + (int64_t)un * a.k + kc, acc[nb][e]);
    }
  }
  if (a.want_bias && (SWAP ? blockIdx.y == 0 && wave == 0 : blockIdx.z == 0)) {
#pragma unroll
    for (int i = 0; i < (SWAP ? NT : 1); ++i) {
      float b = bsum[i];
      b += __shfl_xor(b, 32, 64);
      const int un = SWAP ? n0 + 32 * i + r : n0 + 32 * wave + r;
      if (h == 0 && un < a.n) ctr_stg(out + (int64_t)a.n * a.k + un, b);
    }
  }
}

}  // namespace

bool ctr_gemm_dlds_dw_ok(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy,
                         const float* gw, int64_t ldgw, int64_t m, int n, int k, int act) {
  if (!gw || ldgw != k || m < 4096) return false;
  if (!((n >= 96 && k >= 32) || (k >= 96 && n >= 32))) return false;
  return act == CTR_ACT_NONE || y != nullptr;
}

int ctr_gemm_dlds_dw(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy, float* gw,
                     float* gb, int64_t m, int n, int k, int act, float* workspace, int64_t workspace_floats,
                     hipStream_t st) {
  // the 128-wide side goes to whichever of (units, inputs) pads less
  auto padded = [](int64_t wide, int64_t narrow) {
    const int nt = narrow <= 32 ? 1 : (narrow <= 64 ? 2 : 4);
    return ctr_ceil_div(wide, kAW) * kAW * ctr_ceil_div(narrow, 32 * nt) * 32 * nt;
  };
  const bool swap = n < 96 || (k >= 96 && padded(k, n) < padded(n, k));
  const int wide = swap ? k : n, narrow = swap ? n : k;
  const int nt = narrow <= 32 ? 1 : (narrow <= 64 ? 2 : 4);
  const int64_t ty = ctr_ceil_div(wide, kAW), tz = ctr_ceil_div(narrow, 32 * nt);
  const int64_t slab = (int64_t)n * k + (gb ? n : 0);
  // one round of the 2 resident workgroups per CU: more, shorter row ranges only add slab traffic (measured:
  // 65536 x 256 x 512 takes 200 us with 512 workgroups, 221-228 us with 768-1536)
  // (rounded down: 540 workgroups would run as a full round plus a round of 28)
  int64_t parts = 256 * 2 / (ty * tz);
  if (parts < 1) parts = 1;
  if (parts * slab > workspace_floats) parts = workspace_floats / slab;
  if (parts < 1) return CTR_ELIMIT;
  int64_t rows = ctr_ceil_div(ctr_ceil_div(m, parts), kBK) * kBK;
  if (rows < 8 * kBK) rows = 8 * kBK;
  parts = ctr_ceil_div(m, rows);
  CTR_REQUIRE(ty <= 65535 && tz <= 65535, CTR_ELIMIT);
  const DwArgs a{gy, ldgy, act == CTR_ACT_NONE ? nullptr : y, ldy, x, ldx, m, n, k, act, rows, workspace, slab, gb ? 1 : 0};
  const dim3 grid((unsigned)parts, (unsigned)ty, (unsigned)tz);
#define CTR_DW(NT_, ACT_, SW_) hipLaunchKernelGGL((gemm_dw_dlds_kernel<NT_, ACT_, SW_>), grid, dim3(kThreads), 0, st, a)
#define CTR_DW_ACT(NT_, SW_)                                      \
  do {                                                            \
    if (act == CTR_ACT_NONE) CTR_DW(NT_, CTR_ACT_NONE, SW_);      \
    else if (act == CTR_ACT_RELU) CTR_DW(NT_, CTR_ACT_RELU, SW_); \
    else CTR_DW(NT_, CTR_ACT_SIGMOID, SW_);                       \
  } while (0)
#define CTR_DW_NT(SW_)            \
  do {                            \
    if (nt == 1) CTR_DW_ACT(1, SW_);      \
    else if (nt == 2) CTR_DW_ACT(2, SW_); \
    else CTR_DW_ACT(4, SW_);              \
  } while (0)
  if (swap) CTR_DW_NT(true);
  else CTR_DW_NT(false);
#undef CTR_DW_NT
#undef CTR_DW_ACT
#undef CTR_DW
  int rc = ctr_launch_status();
  if (rc != CTR_OK) return rc;
  CtrSegments segs;
  segs.n = 0;
  segs.s[segs.n++] = CtrSegment{0, (int64_t)n * k, gw};
  if (gb) segs.s[segs.n++] = CtrSegment{(int64_t)n * k, n, gb};
  return ctr_reduce_segments(workspace, (int)parts, slab, segs, st);
}
