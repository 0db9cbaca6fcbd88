// dst[e] += sum_p ws[p*stride + off + e]: the second pass of every "many workgroups
// reduce into a few small tensors" step (weight/bias gradients, bag-table gradients).
// The first pass stores per-workgroup partials with plain stores; summing them here in
// a fixed order replaces chains of same-address fp32 atomics, which the memory side
// serialises (~60 ns per add: 1024 workgroups adding to one weight = 60 us), and makes
// these gradients bitwise reproducible.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;

// workgroup = 16 consecutive outputs x 16 part-lanes: lane (o, q) sums every 16th partial of output
// o with 8 independent loads in flight (256 partials = two rounds of latency; with 4 part-lanes
// the 64 loads per lane were eight dependent rounds and this pass took 7 us for 12 MB).  A wave
// holds 4 part-lanes of each output (xor-shuffles 16 and 32), the 4 waves meet in LDS; the order
// of the sum is fixed.
constexpr int kOut = 16, kPl = kBlock / kOut;

__global__ void __launch_bounds__(kBlock)
reduce_segments_kernel(const float* __restrict__ ws, int parts, int64_t stride, const CtrSegments segs) {
  __shared__ float s_part[kBlock / 64][kOut];
  const CtrSegment sg = segs.s[blockIdx.y];
  const int o = threadIdx.x % kOut, pl = threadIdx.x / kOut, wave = threadIdx.x >> 6;
  for (int64_t e0 = (int64_t)blockIdx.x * kOut; e0 < sg.count; e0 += (int64_t)gridDim.x * kOut) {
    const int64_t e = e0 + o;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
    if (e < sg.count) {
      const float* src = ws + sg.off + e;
      int p = pl;
      for (; p + 7 * kPl < parts; p += 8 * kPl) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += src[(int64_t)(p + kPl * u) * stride];
      }
      for (; p < parts; p += kPl) acc[0] += src[(int64_t)p * stride];
    }
    float t = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if ((threadIdx.x & 63) < kOut) s_part[wave][o] = t;
    __syncthreads();
    if (threadIdx.x < kOut && e < sg.count)
      sg.dst[e] += (s_part[0][o] + s_part[1][o]) + (s_part[2][o] + s_part[3][o]);
    __syncthreads();
  }
}

}  // namespace

int ctr_reduce_segments(const float* ws, int parts, int64_t stride, const CtrSegments& segs, hipStream_t st) {
  if (segs.n == 0 || parts == 0) return CTR_OK;
  int64_t longest = 0;
  for (int i = 0; i < segs.n; ++i) longest = segs.s[i].count > longest ? segs.s[i].count : longest;
  int64_t gx = ctr_ceil_div(longest, kOut);
  if (gx > 2048) gx = 2048;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(reduce_segments_kernel, dim3((unsigned)gx, (unsigned)segs.n), dim3(kBlock), 0, st, ws, parts, stride,
                     segs);
  return ctr_launch_status();
}
