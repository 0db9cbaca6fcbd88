// Y = act(X W^T + b) for WIDE layers on a long batch, with a 256 x 256 macro tile: one 256-thread workgroup per CU, four
// waves as 2 x 2, each wave 128 x 128 outputs = 4 x 4 blocks of v_mfma_f32_32x32x2_f32 (256 accumulator registers).
//
// Why a second forward kernel (profiles/r03_gemm_counters.txt): gemm_dlds.hip's 128 x 128 workgroup tile gives a wave
// 32 x 128 outputs, so a 16-deep step costs it 10 ds_read_b128 and a barrier for 32 MFMAs, every A tile is fetched
// once per 128 output columns, and three workgroups share a CU; hipBLASLt's kernel for 65536 x 256 x 512 runs a
// 256 x 256 x 32 macro tile at one workgroup per CU and issues 1/8 of the vector, 1/7 of the scalar and 0.6 of the LDS
// instructions for the same MFMA work (91 against 104 TF).  Here a step is 16 ds_read_b128 for 128 MFMAs per wave.
//
// Taken by ctr_linear_fwd when it applies (ctr_gemm_wide_ok): n a multiple of 256, k a multiple of 16 and >= 256, aligned
// operands, a batch of at least 4096 rows, no residual -- and CTR_GEMM_WIDE=1: this first version reaches 96.3 TF on
// 65536 x 256 x 512 in the microbenchmark (gemm_dlds.hip 90.5, hipBLASLt 106.7) but loses inside the models' steps;
// profiles/r03_gemm_wide.txt has both measurements and what it still lacks.  Operands stream global -> LDS with the same 16-byte-chunk
// layout, swizzle and ring of three stages as gemm_dlds.hip.
#include "ctr_common.h"

#include <stdlib.h>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kTM = 256, kTN = 256, kBK = 16;
constexpr int kStages = 3;
constexpr int kStageFloats = (kTM + kTN) * kBK;      // one stage: the A tile, then the B tile
constexpr int kLoads = (kTM + kTN) * 4 / kThreads;   // 16-byte chunks a thread requests per step: 8

struct WideArgs {
  const float* x; int64_t ldx;
  const float* w; int64_t ldw;
  const float* bias;
  float* y; int64_t ldy;
  int64_t m; int n; int k; int act;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ f32x4 lds_read128(uint32_t byte_addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
  return v;
}
__device__ __forceinline__ uint32_t lds_addr(const float* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

__global__ void __launch_bounds__(kThreads, 1)
gemm_wide_fwd_kernel(const WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // kStages * kStageFloats
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int64_t mtiles = (a.m + kTM - 1) / kTM;
  const int ntiles = a.n / kTN;
  const int64_t tiles = mtiles * ntiles;
  const int nk = a.k / kBK;

  // per-lane source rows of the 8 chunks this thread requests per step (rows past the batch re-read the last row:
  // their outputs are not stored).  Chunk slot q = 64 * wave + lane + 256 * i: tile row q / 4, 16-byte chunk
  // (q % 4) ^ ((row / 2) % 4) of the step -- the layout gemm_dlds.hip reads back without bank conflicts.
  const float* src[kLoads];
  auto set_tile = [&](int64_t t) {
    const int64_t i0 = (t / ntiles) * kTM, j0 = (int64_t)(t % ntiles) * kTN;
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
      const int q = 64 * wave + lane + kThreads * i;          // 0 .. 2047: A chunks, then B chunks
      const bool isb = q >= kTM * 4;
      const int qq = isb ? q - kTM * 4 : q;
      const int row = qq >> 2, c = (qq & 3) ^ ((row >> 1) & 3);
      int64_t gr = (isb ? j0 : i0) + row;
      if (!isb) gr = gr < a.m ? gr : a.m - 1;
      src[i] = (isb ? a.w + gr * a.ldw : a.x + gr * a.ldx) + c * 4;
    }
  };
  auto issue = [&](int stage, int k) {
    float* st = lds + stage * kStageFloats;
#pragma unroll
    for (int i = 0; i < kLoads; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (int64_t)k * kBK),
                                       (__attribute__((address_space(3))) void*)(st + (64 * wave + kThreads * i) * 4), 16, 0, 0);
  };
  // fragment byte offsets inside a stage: A rows wr*128 + 32 i + r, B rows wc*128 + 32 j + r (B tile behind the A tile)
  uint32_t aoff[4][2], boff[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wr * 128 + 32 * i + r, rb = wc * 128 + 32 * i + r;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      aoff[i][v] = (uint32_t)(ra * 4 + ((2 * h + v) ^ ((ra >> 1) & 3))) * 16u;
      boff[i][v] = (uint32_t)(kTM * kBK * 4) + (uint32_t)(rb * 4 + ((2 * h + v) ^ ((rb >> 1) & 3))) * 16u;
    }
  }
  const uint32_t base0 = lds_addr(lds);

  // a workgroup walks the column tiles of ITS row tiles back to back: the 256 x k operand rows are then re-read from this
  // XCD's L2 (neighbouring workgroups sit on different XCDs: splitting a row tile's columns over them fetched X from
  // HBM once per column tile -- 65536 x 512 x 416 inside DeepFM-26: 318 us against 278 for the 128-wide kernel)
  for (int64_t tile = (int64_t)blockIdx.x * ntiles; tile < tiles; tile = (tile + 1) % ntiles ? tile + 1 : tile + 1 + ((int64_t)gridDim.x - 1) * ntiles) {
    const int64_t i0 = (tile / ntiles) * kTM, j0 = (int64_t)(tile % ntiles) * kTN;
    floatx16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    set_tile(tile);
    __builtin_amdgcn_s_barrier();              // (everybody is done reading the previous tile's last stages)
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int stage = 0;
    for (int ks = 0; ks < nk; ++ks) {
      // my requests of the stage to multiply have landed (those of the next stage may be in flight), then everybody's
      if (ks + 1 < nk) wait_vmcnt<kLoads>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      f32x4 fa[4][2], fb[4][2];
      const uint32_t sb = base0 + (uint32_t)stage * (kStageFloats * 4);
      // the fragments of the step's first half (kk 0..3 | 8..11), then the next stage's requests, then the second half:
      // one wave per SIMD has nobody to hide its LDS latency behind, so the second half is waited for only after the
      // first 64 MFMAs (LDS reads return in order: lgkmcnt(8) = the first eight have landed)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i][0] = lds_read128(sb + aoff[i][0]);
        fb[i][0] = lds_read128(sb + boff[i][0]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i][1] = lds_read128(sb + aoff[i][1]);
        fb[i][1] = lds_read128(sb + boff[i][1]);
      }
      int refill = stage + 2;
      refill = refill >= kStages ? refill - kStages : refill;
      if (ks + 2 < nk) issue(refill, ks + 2);
      // every fragment register passes through its wait, so no MFMA that uses it can be scheduled above it
      asm volatile("s_waitcnt lgkmcnt(8)"
                   : "+v"(fa[0][0]), "+v"(fa[1][0]), "+v"(fa[2][0]), "+v"(fa[3][0]), "+v"(fb[0][0]), "+v"(fb[1][0]),
                     "+v"(fb[2][0]), "+v"(fb[3][0]));
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][0][t], fb[j][0][t], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);     // (the wait below must stay behind these MFMAs: nothing else ties it there)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0][1]), "+v"(fa[1][1]), "+v"(fa[2][1]), "+v"(fa[3][1]), "+v"(fb[0][1]), "+v"(fb[1][1]),
                     "+v"(fb[2][1]), "+v"(fb[3][1]));
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][1][t], fb[j][1][t], acc[i][j], 0, 0, 0);
      stage = stage + 1 == kStages ? 0 : stage + 1;
    }
    // C/D map of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = j0 + wc * 128 + 32 * j + r;
      const float bj = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row0 = i0 + wr * 128 + 32 * i + 4 * h;
        float* yp = a.y + row0 * a.ldy + col;
        if (row0 + 28 < a.m) {                 // (uniform per half-wave block except in the batch's last tile)
#pragma unroll
          for (int e = 0; e < 16; ++e) ctr_stg(yp + ((e & 3) + 8 * (e >> 2)) * a.ldy, ctr_act(acc[i][j][e] + bj, a.act));
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (row0 + (e & 3) + 8 * (e >> 2) < a.m) ctr_stg(yp + ((e & 3) + 8 * (e >> 2)) * a.ldy, ctr_act(acc[i][j][e] + bj, a.act));
        }
      }
    }
  }
}

}  // namespace

bool ctr_gemm_wide_ok(const float* x, int64_t ldx, const float* w, int64_t ldw, int64_t m, int n, int k) {
  // OFF by default: faster than gemm_dlds.hip in the microbenchmark (operands served by the Infinity Cache) but slower
  // inside the steps, where X was just written by another kernel and comes from HBM (profiles/r03_gemm_wide.txt)
  static const int enabled = [] { const char* e = getenv("CTR_GEMM_WIDE"); return e ? atoi(e) : 0; }();
  // (a deep contraction only: the epilogue -- 256 dword stores per wave at one wave per SIMD -- is what a short one is
  // made of: 65536 x 256 x 128 67 us against 58 us for gemm_dlds.hip, x 512 180 against 189: profiles/r03_gemm_wide.txt)
  return enabled && m >= 4096 && n % kTN == 0 && k % kBK == 0 && k >= 256 && ctr_aligned16(x) && ctr_aligned16(w) &&
         ldx % 4 == 0 && ldw % 4 == 0;
}

int ctr_gemm_wide_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy,
                      int64_t m, int n, int k, int act, hipStream_t st) {
  const WideArgs a{x, ldx, w, ldw, bias, y, ldy, m, n, k, act};
  const int64_t mtiles = ctr_ceil_div(m, kTM);
  const int64_t grid = mtiles < 256 ? mtiles : 256;          // one row tile (all its column tiles) per workgroup and round
  constexpr int lds_bytes = kStages * kStageFloats * (int)sizeof(float);   // 96 KB
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          lds_bytes) != hipSuccess)
    return CTR_ELAUNCH;
  hipLaunchKernelGGL(gemm_wide_fwd_kernel, dim3((unsigned)grid), dim3(kThreads), lds_bytes, st, a);
  return ctr_launch_status();
}
