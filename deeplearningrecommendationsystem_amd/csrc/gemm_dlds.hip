// Forward GEMM  Y = act(X W^T + b (+R))  with operands streamed global -> LDS directly
// (global_load_lds_dwordx4, gfx950): no staging registers, no ds_write, and a THREE-stage LDS
// ring, so the loads of pipeline slot g+2 are issued at step g and have two steps to land
// (the register-staged tile kernel of linear.hip has one; PMC: 43 % of its wave time in vmcnt).
//
// Tile 128 x (32*NT) per 256-thread workgroup, 16-deep steps, persistent over the M tiles like
// linear.hip.  A direct load writes lane L's 16 bytes at M0 + 16*L: the LDS image of a wave's
// instruction is one contiguous 1 KB run and cannot be padded, so bank conflicts are avoided by
// choosing WHICH global chunk each lane fetches: slot q of a stage (16 B each, row = q/4) holds
// k-chunk (q & 3) ^ ((row >> 1) & 3) of its row -- the 8 rows a quarter-wave reads with one
// ds_read_b128 then cover all 32 banks.
// The 16-byte direct loads only need 4-byte aligned addresses (measured: rows with an odd leading
// dimension run at the same rate), so any X / W layout with K >= 4 is taken.  Rows past M / N are
// clamped to the last row: their products are computed and dropped.  Contraction tail (K % 16 != 0):
// a chunk that would cross the end of its row is fetched from K-4 instead, so no load ever leaves
// the matrix; the X fragment zeroes the positions that are duplicates or past K, W is left as is
// (finite, multiplied by zero).
#include "ctr_common.h"

#include <type_traits>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kBM = 128;
constexpr int kBK = 16;
constexpr int kStages = 3;

struct DldsArgs {
  const float* x; int64_t ldx;
  const float* w; int64_t ldw;
  const float* bias;
  const float* res; int64_t ldr;
  float* y; int64_t ldy;
  int64_t m; int n; int64_t k; int act;
  int res_group;        // > 1: residual row of output row i is i / res_group (one row per group of consecutive rows)
  CtrFastDiv res_div;
  uint32_t* mask; int64_t ldmask;  // optional: bit (j & 31) of mask[i*ldmask + j/32] = (Y[i,j] > 0); needs n % 32 == 0
  // optional single-unit layer on top (n <= 32*NT: the workgroup holds whole rows): dot_out[i] = Y[i,:] . dot_w + dot_b[0]
  const float* dot_w; const float* dot_b; float* dot_out; int64_t lddot;
};

// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}

// fetch one operand tile (ROWS x 16 floats) into a stage: chunk slot q = 64*wave + lane + 256*i
template <int ROWS>
__device__ __forceinline__ void fetch(float* stage, const float* __restrict__ src, int64_t ld, int64_t row0,
                                      int64_t rows_total, int64_t k0, int64_t k_total, int lane, int wave) {
  constexpr int kChunks = ROWS * 4;
  constexpr int kIters = (kChunks + kThreads - 1) / kThreads;
#pragma unroll
  for (int i = 0; i < kIters; ++i) {
    // first slot of this wave's instruction (uniform).  A tile of fewer chunks than threads is fetched twice
    // (same bytes to the same slots): every wave then has the same number of loads in flight, which
    // the vmcnt bookkeeping of the ring relies on
    int q0 = 64 * wave + kThreads * i;
    if (kChunks % kThreads != 0 && q0 >= kChunks) q0 -= kChunks;
    const int q = q0 + lane;
    const int row = q >> 2, c = (q & 3) ^ ((row >> 1) & 3);
    int64_t gr = row0 + row;
    gr = gr < rows_total ? gr : rows_total - 1;
    int64_t kc = k0 + c * 4;
    kc = kc < k_total - 4 ? kc : k_total - 4;
    const float* g = src + gr * ld + kc;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(stage + q0 * 4), 16, 0, 0);
  }
}

// The same fetch with the per-lane part of the address (row clamp, chunk swizzle, 64-bit row offset: ~12 vector
// instructions per load) formed ONCE per tile: a 16-deep step that lies wholly inside K then costs one 64-bit add per
// load.  (SQ counters of 65536 x 256 x 512: 79 non-MFMA vector + 50 scalar instructions per step and wave next to 32
// MFMAs, against ~10 in hipBLASLt's kernel -- profiles/r03_gemm_counters.txt.)
template <int ROWS>
struct TileLanes {
  static constexpr int kIters = (ROWS * 4 + kThreads - 1) / kThreads;
  const float* p[kIters];
};
template <int ROWS>
__device__ __forceinline__ void tile_lanes(TileLanes<ROWS>& t, const float* __restrict__ src, int64_t ld, int64_t row0,
                                           int64_t rows_total, int lane, int wave) {
  constexpr int kChunks = ROWS * 4;
#pragma unroll
  for (int i = 0; i < TileLanes<ROWS>::kIters; ++i) {
    int q0 = 64 * wave + kThreads * i;
    if (kChunks % kThreads != 0 && q0 >= kChunks) q0 -= kChunks;
    const int q = q0 + lane;
    const int row = q >> 2, c = (q & 3) ^ ((row >> 1) & 3);
    int64_t gr = row0 + row;
    gr = gr < rows_total ? gr : rows_total - 1;
    t.p[i] = src + gr * ld + c * 4;
  }
}
template <int ROWS>
__device__ __forceinline__ void fetch_inside(float* stage, const TileLanes<ROWS>& t, int64_t k0, int wave) {
  constexpr int kChunks = ROWS * 4;
#pragma unroll
  for (int i = 0; i < TileLanes<ROWS>::kIters; ++i) {
    int q0 = 64 * wave + kThreads * i;
    if (kChunks % kThreads != 0 && q0 >= kChunks) q0 -= kChunks;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t.p[i] + k0),
                                     (__attribute__((address_space(3))) void*)(stage + q0 * 4), 16, 0, 0);
  }
}

// The operand fragments are read with ds_read_b128 written as asm: the compiler cannot tell which
// stage an LDS-DMA load targets and would put s_waitcnt vmcnt(0) in front of every ordinary LDS read --
// waiting for the loads issued a moment ago, i.e. no pipeline at all.  The waits are placed by hand
// (wait_vmcnt before the barrier, lds_fence after the reads).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 lds_read128(uint32_t byte_addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
  return v;
}
__device__ __forceinline__ uint32_t lds_addr(const float* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}
// byte offset inside a stage of the two 16-byte chunks (kk = 8h .. 8h+3, 8h+4 .. 8h+7) of tile row `row`
__device__ __forceinline__ void frag_offsets(int row, int h, uint32_t (&off)[2]) {
  const int sw = (row >> 1) & 3;
#pragma unroll
  for (int v = 0; v < 2; ++v) off[v] = (uint32_t)(row * 4 + ((2 * h + v) ^ sw)) * 16u;
}

template <int NT>
__global__ void __launch_bounds__(kThreads, 3)
gemm_fwd_dlds_kernel(const DldsArgs a) {
  constexpr int BN = 32 * NT;
  constexpr int kLoadsA = kBM * 4 / kThreads;                      // 2
  constexpr int kLoadsB = (BN * 4 + kThreads - 1) / kThreads;      // NT=4: 2, NT=2: 1, NT=1: 1 (half the waves)
  constexpr int kPerSlot = kLoadsA + kLoadsB;
  __shared__ __attribute__((aligned(16))) float s_a[kStages][kBM * kBK];
  __shared__ __attribute__((aligned(16))) float s_b[kStages][BN * kBK];

  const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t mtiles = (a.m + kBM - 1) / kBM;
  const int64_t j0 = (int64_t)blockIdx.y * BN;
  const int nk = (int)((a.k + kBK - 1) / kBK);
  const bool ktail = a.k % kBK != 0;
  int64_t tile = blockIdx.x;
  if (tile >= mtiles) return;

  constexpr int CH = 4 / NT;
  floatx16 acc[NT][CH];
  // the workgroup's bias columns, fetched once: a load inside the tile loop's epilogue would have to
  // drain the in-order vmcnt queue, i.e. wait for the operand loads of the next tile
  float bj[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int64_t j = j0 + 32 * n + (lane0 & 31);
    bj[n] = a.bias && j < a.n ? a.bias[j] : 0.0f;
  }
  float dwj[NT];   // the head's weights of this lane's columns (zero past n: those columns add nothing)
  float dotb = 0.0f;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int64_t j = j0 + 32 * n + (lane0 & 31);
    dwj[n] = a.dot_out && j < a.n ? a.dot_w[j] : 0.0f;
  }
  if (a.dot_out && a.dot_b) dotb = a.dot_b[0];
  wait_vmcnt<0>();  // ... and landed before the ring starts, so that no later use of bj waits on the queue

  auto advance = [&](int64_t& t, int& k) {
    if (++k == nk) {
      k = 0;
      t += gridDim.x;
    }
  };
  TileLanes<kBM> la;
  TileLanes<BN> lb;
  int64_t la_tile = -1;
  tile_lanes<BN>(lb, a.w, a.ldw, j0, a.n, lane0, wave);     // the workgroup's weight rows never change
  auto issue = [&](int stage, int64_t t, int k) {
    const int64_t k0 = (int64_t)k * kBK;
    if (k0 + kBK <= a.k) {                                    // (uniform) the whole step lies inside K
      if (t != la_tile) {
        tile_lanes<kBM>(la, a.x, a.ldx, t * kBM, a.m, lane0, wave);
        la_tile = t;
      }
      fetch_inside<kBM>(s_a[stage], la, k0, wave);
      fetch_inside<BN>(s_b[stage], lb, k0, wave);
    } else {
      fetch<kBM>(s_a[stage], a.x, a.ldx, t * kBM, a.m, k0, a.k, lane0, wave);
      fetch<BN>(s_b[stage], a.w, a.ldw, j0, a.n, k0, a.k, lane0, wave);
    }
  };
  // slots g+1 and g+2 relative to the one being multiplied
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
    asm volatile("" : "+v"(lane));  // keep the epilogue's 64 addresses out of the tile loop's live set
    const int r = lane & 31, h = lane >> 5;
    uint32_t aoff[2], boff[2];
    frag_offsets(32 * wave + r, h, aoff);
    frag_offsets(r, h, boff);  // rows 32n + r: same swizzle for every n (32n >> 1 is a multiple of 4)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][c][e] = 0.0f;
    // grouped residual (DIN: u[b] for the L rows of sample b): the wave's 32 rows lie in at most two groups, so a
    // lane needs two values per column tile.  Requested here, in front of the contraction: a load issued in the
    // epilogue is waited for together with the stores of the column tiles before it.
    const bool grouped = a.res && a.res_group >= 32;
    const int64_t first = i0 + 32 * wave;
    int64_t edge = 0;
    float r0v[NT], r1v[NT];
    if (grouped) {
      const int64_t g0 = (int64_t)ctr_div((uint32_t)(first < a.m ? first : a.m - 1), a.res_div);
      edge = (g0 + 1) * a.res_group;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int64_t j = j0 + 32 * n + r;
        const int64_t jc = j < a.n ? j : a.n - 1;
        r0v[n] = ctr_ldg(a.res + g0 * a.ldr + jc);
        r1v[n] = ctr_ldg(a.res + (edge < a.m ? g0 + 1 : g0) * a.ldr + jc);
      }
    }
    for (int ks = 0; ks < nk; ++ks) {
      // my loads of the slot to multiply have landed (only those of the next slot may be in flight),
      // then everybody's have, and everybody is done reading the stage that is refilled next
      if (t1 < mtiles) wait_vmcnt<kPerSlot>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      int refill = stage + 2;
      refill = refill >= kStages ? refill - kStages : refill;
      f32x4 fa[2], fb[NT][2];
      const uint32_t abase = lds_addr(s_a[0]) + (uint32_t)stage * (kBM * kBK * 4);
      const uint32_t bbase = lds_addr(s_b[0]) + (uint32_t)stage * (BN * kBK * 4);
#pragma unroll
      for (int v = 0; v < 2; ++v) fa[v] = lds_read128(abase + aoff[v]);
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int v = 0; v < 2; ++v) fb[n][v] = lds_read128(bbase + boff[v] + (uint32_t)n * (32 * kBK * 4));
      if (t2 < mtiles) issue(refill, t2, k2);
      advance(t1, k1);
      advance(t2, k2);
      // every fragment register passes through the wait, so no MFMA can be scheduled above it
      if constexpr (NT == 1)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0][0]), "+v"(fb[0][1]));
      else if constexpr (NT == 2)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]),
                       "+v"(fb[2][0]), "+v"(fb[2][1]), "+v"(fb[3][0]), "+v"(fb[3][1]));
      if (ktail && ks == nk - 1) {
        // chunk (2h + v) of the last step starts at kc; fetched from K-4 when it crosses K: its first
        // kc + 4 - K positions then repeat the previous chunk (or everything is past K)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int64_t over = (int64_t)ks * kBK + (2 * h + v) * 4 + 4 - a.k;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < over) fa[v][j] = 0.0f;
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[n][t % CH] =
              __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t >> 2][t & 3], fb[n][t >> 2][t & 3], acc[n][t % CH], 0, 0, 0);
      stage = stage + 1 == kStages ? 0 : stage + 1;
    }
    // C/D map of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float pdot[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) pdot[e] = 0.0f;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int64_t j = j0 + 32 * n + r;
      if (j < a.n) {
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          v[e] = acc[n][0][e] + bj[n];
#pragma unroll
          for (int c = 1; c < CH; ++c) v[e] += acc[n][c][e];
        }
        if (grouped) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int64_t i = first + (e & 3) + 8 * (e >> 2) + 4 * h;
            v[e] += i < edge ? r0v[n] : r1v[n];
          }
        } else if (a.res) {
          // all 16 residual values in flight at once (the wait for them also drains the ring's loads)
          float rv[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int64_t i = i0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int64_t ri = a.res_group > 1 ? (int64_t)ctr_div((uint32_t)(i < a.m ? i : 0), a.res_div) : i;
            rv[e] = i < a.m ? ctr_ldg(a.res + ri * a.ldr + j) : 0.0f;
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] += rv[e];
        }
        // sign bits of the outputs for the backward (1 bit instead of 4 bytes per element re-read there): a ballot
        // per register gives the 32 columns of two rows; lane r of the first half-wave keeps the word of row r
        // (register (r & 3) + 4 * (r >> 3), upper half of the ballot when r & 4).  n % 32 == 0: every lane is here.
        uint32_t word = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t i = i0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * h;
          const float o = ctr_act(v[e], a.act);
          if (i < a.m) ctr_stg(a.y + i * a.ldy + j, o);
          pdot[e] = fmaf(o, dwj[n], pdot[e]);
          if (a.mask) {
            const uint64_t bal = __ballot(o > 0.0f);
            const uint32_t cand = (r & 4) ? (uint32_t)(bal >> 32) : (uint32_t)bal;
            if (e == (r & 3) + 4 * (r >> 3)) word = cand;
          }
        }
        if (a.mask && h == 0 && i0 + 32 * wave + r < a.m)
          a.mask[(i0 + 32 * wave + r) * a.ldmask + ((j0 + 32 * n) >> 5)] = word;
      }
    }
    if (a.dot_out) {
      // transposed reduction over the 32 columns a half-wave holds: lanes r and r ^ s swap half of their values,
      // so after four stages lane r holds register r >> 1, summed over 16 lanes, and one more exchange finishes it
      // (16 shuffles for 16 rows instead of 16 five-step reductions)
      auto stage = [&](auto keepv, int bit, int xr) __attribute__((always_inline)) {
        constexpr int KEEP = decltype(keepv)::value;
#pragma unroll
        for (int q = 0; q < KEEP; ++q) {
          const float send = bit ? pdot[q] : pdot[q + KEEP];
          const float keep = bit ? pdot[q + KEEP] : pdot[q];
          pdot[q] = keep + __shfl_xor(send, xr, 64);
        }
      };
      stage(std::integral_constant<int, 8>{}, r & 16, 16);
      stage(std::integral_constant<int, 4>{}, r & 8, 8);
      stage(std::integral_constant<int, 2>{}, r & 4, 4);
      stage(std::integral_constant<int, 1>{}, r & 2, 2);
      const float t = pdot[0] + __shfl_xor(pdot[0], 1, 64);
      const int e = r >> 1;
      const int64_t i = i0 + 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * h;
      if ((r & 1) == 0 && i < a.m) a.dot_out[i * a.lddot] = t + dotb;
    }
  }
}

}  // namespace

bool ctr_gemm_dlds_ok(const float* x, int64_t ldx, const float* w, int64_t ldw, int64_t m, int n, int k) {
  return k >= 4 && m >= 1 && n >= 1;  // a chunk is fetched from k-4 at the latest
}

static int launch_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* res,
                      int64_t ldr, float* y, int64_t ldy, int64_t m, int n, int k, int act, hipStream_t st,
                      int res_group = 1, uint32_t* mask = nullptr, int64_t ldmask = 0, const float* dot_w = nullptr,
                      const float* dot_b = nullptr, float* dot_out = nullptr, int64_t lddot = 0) {
  int nt = n <= 32 ? 1 : (n <= 64 ? 2 : 4);
  const int64_t mtiles = ctr_ceil_div(m, kBM);
  // few rows (a table of ~1000 rows instead of a batch): 128 x 128 tiles would leave most CUs without a workgroup
  // (943 x 256: 16 of them) -- narrower column tiles re-read the few rows from L2 and fill the chip
  while (nt > 1 && !dot_out && mtiles * ctr_ceil_div(n, 32 * nt) < 128) nt >>= 1;
  const int64_t ny = ctr_ceil_div(n, 32 * nt);
  int64_t gx = 256 * 3 / ny;  // rounded down: a workgroup beyond the resident 3 per CU would start a second round
  if (gx > mtiles) gx = mtiles;
  if (gx < 1) gx = 1;
  CTR_REQUIRE(ny <= 65535, CTR_ELIMIT);
  CTR_REQUIRE(res_group >= 1 && (res_group == 1 || m < (1ll << 32)), CTR_ELIMIT);
  CTR_REQUIRE(!mask || (n % 32 == 0 && ldmask >= n / 32), CTR_EINVAL);
  const DldsArgs a{x, ldx, w, ldw, bias, res, ldr, y, ldy, m, n, (int64_t)k, act, res_group,
                   ctr_fastdiv((uint32_t)res_group), mask, ldmask, dot_w, dot_b, dot_out, lddot};
  CTR_REQUIRE(!dot_out || (ny == 1 && dot_w), CTR_EINVAL);  // the head needs whole rows in one workgroup
  const dim3 grid((unsigned)gx, (unsigned)ny);
  if (nt == 1) hipLaunchKernelGGL(gemm_fwd_dlds_kernel<1>, grid, dim3(kThreads), 0, st, a);
  else if (nt == 2) hipLaunchKernelGGL(gemm_fwd_dlds_kernel<2>, grid, dim3(kThreads), 0, st, a);
  else hipLaunchKernelGGL(gemm_fwd_dlds_kernel<4>, grid, dim3(kThreads), 0, st, a);
  return ctr_launch_status();
}

// Y = act(X W^T + b + R[row / group]): one residual row per `group` consecutive output rows
int ctr_gemm_dlds_fwd_group(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* res,
                            int64_t ldr, int group, float* y, int64_t ldy, int64_t m, int n, int k, int act,
                            hipStream_t st, uint32_t* mask, int64_t ldmask) {
  return launch_fwd(x, ldx, w, ldw, bias, res, ldr, y, ldy, m, n, k, act, st, group, mask, ldmask);
}

int ctr_gemm_dlds_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* res,
                      int64_t ldr, float* y, int64_t ldy, int64_t m, int n, int k, int act, hipStream_t st) {
  // 17..64 units past a multiple of 128 (161 = 128 + 33) would cost a whole, mostly empty 128-wide column
  // of tiles: they get their own launch with a 64-wide tile (91 -> 79 us on 65536 x 161 x 256).  Fewer
  // are not worth a second pass over X (641 = 5 * 128 + 1: 707 us in one launch, 750 split -- with a
  // 32-wide tile or with the single-unit streaming kernel for the odd unit)
  const int rem = n % 128;
  if (n > 128 && rem > 16 && rem <= 64) {
    const int main_n = n - rem;
    int rc = launch_fwd(x, ldx, w, ldw, bias, res, ldr, y, ldy, m, main_n, k, act, st);
    if (rc != CTR_OK) return rc;
    return launch_fwd(x, ldx, w + (int64_t)main_n * ldw, ldw, bias ? bias + main_n : nullptr, res ? res + main_n : nullptr,
                      ldr, y + main_n, ldy, m, rem, k, act, st);
  }
  return launch_fwd(x, ldx, w, ldw, bias, res, ldr, y, ldy, m, n, k, act, st);
}

// C ABI (include/ctrhip.h): Y = act(X W^T + b + R[row / group]);  optional sign bits of Y for the backward
extern "C" int ctr_linear_group_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                                    const float* res, int64_t ldr, int group, float* y, int64_t ldy,
                                    uint32_t* mask /*nullable*/, int64_t ldmask, int64_t m, int n, int k, int act,
                                    void* stream) {
  CTR_REQUIRE(m >= 0 && n >= 1 && k >= 1, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  CTR_REQUIRE(x && w && res && y && group >= 1 && ldx >= k && ldw >= k && ldy >= n && ldr >= n, CTR_EINVAL);
  CTR_REQUIRE(act >= CTR_ACT_NONE && act <= CTR_ACT_SIGMOID, CTR_EINVAL);
  CTR_REQUIRE(ctr_gemm_dlds_ok(x, ldx, w, ldw, m, n, k) && n <= 128, CTR_ELIMIT);  // one launch: no column split
  CTR_REQUIRE(!mask || (n % 32 == 0 && ldmask >= n / 32), CTR_EINVAL);
  return ctr_gemm_dlds_fwd_group(x, ldx, w, ldw, bias, res, ldr, group, y, ldy, m, n, k, act, (hipStream_t)stream, mask,
                                 ldmask);
}

// C ABI (include/ctrhip.h): Y = act(X W^T + b), and a single-unit layer on top of it in the same pass:
// out[i] = Y[i, :] . u + c[0]   (DIN: attention layer 2 and the score layer, model/din.py:45-46)
extern "C" int ctr_linear_fwd_dot(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y,
                                  int64_t ldy, const float* u, const float* c, float* out, int64_t ldout, int64_t m,
                                  int n, int k, int act, void* stream) {
  CTR_REQUIRE(m >= 0 && n >= 1 && k >= 1, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  CTR_REQUIRE(x && w && y && u && out && ldx >= k && ldw >= k && ldy >= n && ldout >= 1, CTR_EINVAL);
  CTR_REQUIRE(act >= CTR_ACT_NONE && act <= CTR_ACT_SIGMOID, CTR_EINVAL);
  CTR_REQUIRE(ctr_gemm_dlds_ok(x, ldx, w, ldw, m, n, k) && n <= 128, CTR_ELIMIT);
  return launch_fwd(x, ldx, w, ldw, bias, nullptr, 0, y, ldy, m, n, k, act, (hipStream_t)stream, 1, nullptr, 0, u, c, out,
                    ldout);
}
