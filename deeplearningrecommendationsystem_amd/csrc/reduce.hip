// dst[e] += sum_p ws[p*stride + off + e]: the second pass of every "many workgroups
// reduce into a few small tensors" step (weight/bias gradients, bag-table gradients).
// The first pass stores per-workgroup partials with plain stores; summing them here in
// a fixed order replaces chains of same-address fp32 atomics, which the memory side
// serialises (~60 ns per add: 1024 workgroups adding to one weight = 60 us), and makes
// these gradients bitwise reproducible.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 512;

// workgroup = 16 consecutive outputs x 16 part-lanes: lane (o, q) sums every 16th partial of output
// o with 8 independent loads in flight (256 partials = two rounds of latency; with 4 part-lanes
// the 64 loads per lane were eight dependent rounds and this pass took 7 us for 12 MB).  A wave
// holds 4 part-lanes of each output (xor-shuffles 16 and 32), the 4 waves meet in LDS; the order
// of the sum is fixed.
constexpr int kOut = 32, kPl = kBlock / kOut;

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
    t += __shfl_xor(t, 32, 64);   // a wave holds 2 part-lanes of each of its 32 outputs
    if ((threadIdx.x & 63) < kOut) s_part[wave][o] = t;
    __syncthreads();
    if (threadIdx.x < kOut && e < sg.count)
      sg.dst[e] += ((s_part[0][o] + s_part[1][o]) + (s_part[2][o] + s_part[3][o])) +
                   ((s_part[4][o] + s_part[5][o]) + (s_part[6][o] + s_part[7][o]));
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256)
zero_fill_kernel(float* __restrict__ p, int64_t n) {   // n % 4 == 0, p 16-byte aligned
  const ctr_f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024)
    *reinterpret_cast<ctr_f32x4*>(p + i) = z;
}

// The head's 73 sums (72 folded weights + the folded bias) of NeuralCF and the chain rule through ctr_fold_head_fwd's
// map in the SAME launch as the other segments (one extra row of the grid, one workgroup): ctr_fold_head_bwd was a
// 4.4 us launch of its own behind this one.  gw / gc receive the sums (+=) as any segment's dst would; the fold
// gradients are formed from the totals.
__global__ void __launch_bounds__(kBlock)
reduce_segments_fold_kernel(const float* __restrict__ ws, int parts, int64_t stride, const CtrSegments segs, int64_t hoff,
                            const CtrHeadFoldGrad F) {
  if (blockIdx.y < (unsigned)segs.n) {
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
      t += __shfl_xor(t, 32, 64);
      if ((threadIdx.x & 63) < kOut) s_part[wave][o] = t;
      __syncthreads();
      if (threadIdx.x < kOut && e < sg.count)
        sg.dst[e] += ((s_part[0][o] + s_part[1][o]) + (s_part[2][o] + s_part[3][o])) +
                     ((s_part[4][o] + s_part[5][o]) + (s_part[6][o] + s_part[7][o]));
      __syncthreads();
    }
    return;
  }
  if (blockIdx.x != 0) return;
  // ---- the head: 128 output lanes (73 used) x 4 part-lanes, then the fold's chain rule (p = 64, n = 64, k = 8)
  constexpr int kHP = 64, kHN = 64, kHK = 8, kSums = kHP + kHK + 1;
  __shared__ float s_h[4][128];
  __shared__ float s_g[kSums];
  const int o = threadIdx.x & 127, pl = threadIdx.x >> 7;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
  if (o < kSums) {
    // this workgroup is alone with 73 x parts loads: sixteen in flight per thread (four rounds for 256 partials)
    const float* src = ws + hoff + o;
    int p = pl;
    for (; p + 15 * 4 < parts; p += 16 * 4) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = src[(int64_t)(p + 4 * u) * stride];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 7] += v[u];
    }
    for (; p < parts; p += 4) acc[0] += src[(int64_t)p * stride];
  }
  s_h[pl][o] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (threadIdx.x < kSums) {
    const float t = (s_h[0][threadIdx.x] + s_h[1][threadIdx.x]) + (s_h[2][threadIdx.x] + s_h[3][threadIdx.x]);
    float* dst = threadIdx.x < kHP + kHK ? F.gwfold + threadIdx.x : F.gcfold;
    const float total = dst[0] + t;
    dst[0] = total;
    s_g[threadIdx.x] = total;
  }
  __syncthreads();
  const float* u = F.u_full + kHP;
  const float gc = s_g[kHP + kHK];
  // gW[i][q] += u[i] * gv[q]: 512 products, one per thread
  {
    const int i = threadIdx.x / kHK, q = threadIdx.x % kHK;
    if (F.gw) F.gw[(int64_t)i * F.ldgw + q] += u[i] * s_g[kHP + q];
  }
  if (threadIdx.x < kHP) {
    if (F.gu_full) F.gu_full[threadIdx.x] += s_g[threadIdx.x];
  } else if (threadIdx.x < kHP + kHN) {
    const int i = threadIdx.x - kHP;
    if (F.gu_full) {
      float sacc = F.b ? F.b[i] * gc : 0.0f;
#pragma unroll
      for (int q = 0; q < kHK; ++q) sacc = fmaf(F.w[(int64_t)i * F.ldw + q], s_g[kHP + q], sacc);
      F.gu_full[kHP + i] += sacc;
    }
  } else if (threadIdx.x < kHP + 2 * kHN) {
    const int i = threadIdx.x - kHP - kHN;
    if (F.gb) F.gb[i] += u[i] * gc;
  } else if (threadIdx.x == kHP + 2 * kHN) {
    if (F.gb2) F.gb2[0] += gc;
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

int ctr_reduce_segments_fold(const float* ws, int parts, int64_t stride, const CtrSegments& segs, int64_t head_off,
                             const CtrHeadFoldGrad& fold, hipStream_t st) {
  if (parts == 0) return CTR_OK;
  int64_t longest = kOut;
  for (int i = 0; i < segs.n; ++i) longest = segs.s[i].count > longest ? segs.s[i].count : longest;
  int64_t gx = ctr_ceil_div(longest, kOut);
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(reduce_segments_fold_kernel, dim3((unsigned)gx, (unsigned)segs.n + 1), dim3(kBlock), 0, st, ws, parts,
                     stride, segs, head_off, fold);
  return ctr_launch_status();
}

// a capturable zero fill (a kernel, not hipMemsetAsync: DESIGN.md section 4, hipGraph bullet)
int ctr_zero_fill(float* p, int64_t n, hipStream_t st) {
  if (n <= 0) return CTR_OK;
  CTR_REQUIRE(p && ctr_aligned16(p) && n % 4 == 0, CTR_EALIGN);
  hipLaunchKernelGGL(zero_fill_kernel, dim3(ctr_stream_grid(n / 4, 256)), dim3(256), 0, st, p, n);
  return ctr_launch_status();
}
