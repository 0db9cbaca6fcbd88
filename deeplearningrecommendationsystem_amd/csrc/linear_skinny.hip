// Weight gradient of an nn.Linear with FEW units on BOTH sides over a very long batch -- DIEN's
// GRU projections (E -> 3E = 16 -> 48 over B*L = 3.3M rows, model/dien.py:47,61):
//     gW[n, :k] += sum_m gz[m, n] x[m, :k],   gb[n] += sum_m gz[m, n],   gz = gy * act'(y).
// As a 32x32 MFMA tile this product is 4/5 padding and every 16-row pipeline step moves ~4 KB
// (555 us = 1.5 TB/s).  It is a streaming reduction: a lane owns an output unit n and keeps its
// whole gW row (k <= 32 floats) in registers; a wave walks a contiguous block of rows, four in
// flight: one coalesced load of gz[m, :n] and the x row as uniform SCALAR loads feeding the FMAs
// as SGPR operands (300 us; broadcast vector loads of the x row measured 570 us, and lane-per-row
// variants of the forward / dX products touch 64 cache lines per instruction and lost to the tile
// kernel, so those two stay there).  Waves are summed through LDS, workgroups through the
// workspace in a fixed order (reduce.hip).
// k in {8, 16, 32}, n <= 64, 16-byte aligned x rows, gW contiguous.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;

template <int K>
__global__ void __launch_bounds__(kBlock)
skinny_dw_kernel(const float* __restrict__ gy, int64_t ldgy, const float* __restrict__ yv, int64_t ldy, int act,
                 const float* __restrict__ x, int64_t ldx, int64_t m, int n, int64_t rows_per_wave,
                 float* __restrict__ ws /* [gridDim.x][n*K + n] */) {
  __shared__ float s_part[kBlock / 64][64 * (K + 1)];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t r0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * rows_per_wave;
  const int64_t r1 = r0 + rows_per_wave < m ? r0 + rows_per_wave : m;
  float acc[K], bsum = 0.0f;
#pragma unroll
  for (int j = 0; j < K; ++j) acc[j] = 0.0f;
  const bool live = lane < n;
  for (int64_t row = r0; row < r1; row += 4) {
    // four rows in flight: a load-use loop over single rows would pay one latency per row
    float g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = live && row + u < r1;
      const int64_t rr = row + u < r1 ? row + u : r1 - 1;
      g[u] = ok ? ctr_ldg(gy + rr * ldgy + lane) : 0.0f;
      if (act != CTR_ACT_NONE && ok) g[u] *= ctr_act_grad(ctr_ldg(yv + rr * ldy + lane), act);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t rr = row + u < r1 ? row + u : r1 - 1;
      const float* xr = x + rr * ldx;  // wave-uniform address: scalar loads
      bsum += g[u];
#pragma unroll
      for (int j = 0; j < K; ++j) acc[j] = fmaf(g[u], xr[j], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < K; ++j) s_part[wave][lane * (K + 1) + j] = acc[j];
  s_part[wave][lane * (K + 1) + K] = bsum;
  __syncthreads();
  float* out = ws + (int64_t)blockIdx.x * ((int64_t)n * K + n);
  for (int i = threadIdx.x; i < n * (K + 1); i += kBlock) {
    float t = 0.0f;
#pragma unroll
    for (int wv = 0; wv < kBlock / 64; ++wv) t += s_part[wv][i];
    const int unit = i / (K + 1), j = i - unit * (K + 1);
    if (j < K) out[unit * K + j] = t;
    else out[(int64_t)n * K + unit] = t;
  }
}

}  // namespace

bool ctr_skinny_dw_ok(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy,
                      const float* gw, int64_t ldgw, int64_t m, int n, int k, int act) {
  if (!(k == 8 || k == 16 || k == 32) || n < 1 || n > 64 || m < 65536) return false;
  if (act != CTR_ACT_NONE && !y) return false;
  return gw && ldgw == k && ctr_aligned16(x) && ldx % 4 == 0 && gy && ldgy >= n && (act == CTR_ACT_NONE || ldy >= n);
}

int ctr_skinny_dw(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy, float* gw,
                  float* gb, int64_t m, int n, int k, int act, float* workspace, int64_t workspace_floats,
                  hipStream_t st) {
  const int64_t slab = (int64_t)n * k + n;
  int64_t blocks = 2048;
  if (blocks * slab > workspace_floats) blocks = workspace_floats / slab;
  CTR_REQUIRE(workspace && blocks >= 1, CTR_ELIMIT);
  const int64_t rows_per_wave = ctr_ceil_div(m, blocks * (kBlock / 64));
  blocks = ctr_ceil_div(m, rows_per_wave * (kBlock / 64));
#define CTR_SK(K_)                                                                                                \
  hipLaunchKernelGGL(skinny_dw_kernel<K_>, dim3((unsigned)blocks), dim3(kBlock), 0, st, gy, ldgy, y, ldy, act, x, ldx, \
                     m, n, rows_per_wave, workspace)
  if (k == 8) CTR_SK(8);
  else if (k == 16) CTR_SK(16);
  else CTR_SK(32);
#undef CTR_SK
  int rc = ctr_launch_status();
  if (rc != CTR_OK) return rc;
  CtrSegments segs;
  segs.n = 0;
  segs.s[segs.n++] = CtrSegment{0, (int64_t)n * k, gw};
  if (gb) segs.s[segs.n++] = CtrSegment{(int64_t)n * k, n, gb};
  return ctr_reduce_segments(workspace, (int)blocks, slab, segs, st);
}
