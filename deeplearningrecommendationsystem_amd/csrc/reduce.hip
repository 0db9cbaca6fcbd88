// dst[e] += sum_p ws[p*stride + off + e]: the second pass of every "many workgroups
// reduce into a few small tensors" step (weight/bias gradients, bag-table gradients).
// The first pass stores per-workgroup partials with plain stores; summing them here in
// a fixed order replaces chains of same-address fp32 atomics, which the memory side
// serialises (~60 ns per add: 1024 workgroups adding to one weight = 60 us), and makes
// these gradients bitwise reproducible.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;

// workgroup = 64 consecutive outputs x 4 part-lanes (one wave each); a lane sums every
// 4th partial with 8 independent accumulators in flight, the 4 waves meet in LDS and the
// final sum is taken in a fixed order
__global__ void __launch_bounds__(kBlock)
reduce_segments_kernel(const float* __restrict__ ws, int parts, int64_t stride, const CtrSegments segs) {
  __shared__ float s_part[4][64];
  const CtrSegment sg = segs.s[blockIdx.y];
  const int lane = threadIdx.x & 63, pl = threadIdx.x >> 6;
  for (int64_t e0 = (int64_t)blockIdx.x * 64; e0 < sg.count; e0 += (int64_t)gridDim.x * 64) {
    const int64_t e = e0 + lane;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
    if (e < sg.count) {
      const float* src = ws + sg.off + e;
      int p = pl;
      for (; p + 28 < parts; p += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += src[(int64_t)(p + 4 * u) * stride];
      }
      for (; p < parts; p += 4) acc[0] += src[(int64_t)p * stride];
    }
    s_part[pl][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (pl == 0 && e < sg.count) sg.dst[e] += (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
    __syncthreads();
  }
}

}  // namespace

int ctr_reduce_segments(const float* ws, int parts, int64_t stride, const CtrSegments& segs, hipStream_t st) {
  if (segs.n == 0 || parts == 0) return CTR_OK;
  int64_t longest = 0;
  for (int i = 0; i < segs.n; ++i) longest = segs.s[i].count > longest ? segs.s[i].count : longest;
  int64_t gx = ctr_ceil_div(longest, 64);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(reduce_segments_kernel, dim3((unsigned)gx, (unsigned)segs.n), dim3(kBlock), 0, st, ws, parts, stride,
                     segs);
  return ctr_launch_status();
}
