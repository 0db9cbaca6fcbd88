// Per-row top-k of a score matrix: the ranking step of every recommendation() of the reference
// (model/mf.py:28-35, neuralcf.py:61-72, pnn.py:133-143, din.py:55-66 call torch.topk on the scores of one user's
// candidates).  One workgroup per row.  Order: score descending, NaN first (torch.topk's convention), equal scores by
// ascending index -- a fixed order, where torch leaves ties unspecified.
//
//   n <= kSortMax : the whole row goes to LDS as 64-bit (ordered score bits << 32 | ~index) words, one bitonic sort,
//                   the first k are written (ml-100k: 1682 items, k = all of them in the reference's MF / NeuralCF)
//   n  > kSortMax : (k <= kSortMax) four 8-bit radix passes over the row find the k-th largest key T and how many
//                   entries equal to T belong to the answer; one more pass collects keys > T (any order) and the FIRST
//                   `need` keys == T (index order, by a workgroup prefix count), then the same sort.  The row is read
//                   five times, from L2 after the first (a 1e6-item row is 4 MB).
#include "ctr_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kSortMax = 4096;

__device__ __forceinline__ uint32_t ordered_key(float v) {
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // ascending in the float order; +NaN above +inf
}
__device__ __forceinline__ float key_value(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ unsigned long long entry(uint32_t key, uint32_t idx) {
  return ((unsigned long long)key << 32) | (uint32_t)~idx;          // larger = better score, then SMALLER index
}

// descending bitonic sort of s[0..len) (len a power of two), all threads of the workgroup
__device__ void sort_desc(unsigned long long* s, int len) {
  for (int size = 2; size <= len; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < len / 2; t += kThreads) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned long long a = s[lo], b = s[hi];
        if ((a < b) == desc) {
          s[lo] = b;
          s[hi] = a;
        }
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ int pow2_at_least(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

__global__ void __launch_bounds__(kThreads)
topk_rows_kernel(const float* __restrict__ scores, int64_t row_stride, int64_t col_stride, int64_t n, int k,
                 int64_t* __restrict__ idx_out, float* __restrict__ val_out) {
  __shared__ unsigned long long s_e[kSortMax];
  __shared__ int s_hist[256];
  __shared__ int s_wave[kThreads / 64][2];
  __shared__ uint32_t s_prefix;
  __shared__ int s_need, s_fill, s_eq_base;
  const float* row = scores + (int64_t)blockIdx.x * row_stride;
  int64_t* out = idx_out + (int64_t)blockIdx.x * k;
  float* vout = val_out ? val_out + (int64_t)blockIdx.x * k : nullptr;
  int len;
  if (n <= kSortMax) {
    len = pow2_at_least((int)n);
    for (int i = threadIdx.x; i < len; i += kThreads)
      s_e[i] = i < n ? entry(ordered_key(row[(int64_t)i * col_stride]), (uint32_t)i) : 0ull;
  } else {
    // ---- radix select: after the passes s_prefix is the k-th largest key, s_need how many entries equal to it count
    if (threadIdx.x == 0) {
      s_prefix = 0u;
      s_need = k;
    }
    uint32_t mask = 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
      s_hist[threadIdx.x] = 0;
      __syncthreads();
      const uint32_t prefix = s_prefix;
      for (int64_t i = threadIdx.x; i < n; i += kThreads) {
        const uint32_t key = ordered_key(row[i * col_stride]);
        if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        int need = s_need, b = 255;
        for (; b > 0; --b) {
          if (s_hist[b] >= need) break;
          need -= s_hist[b];
        }
        s_need = need;
        s_prefix = prefix | ((uint32_t)b << shift);
      }
      mask |= 0xffu << shift;
      __syncthreads();
    }
    const uint32_t T = s_prefix;
    const int need = s_need;           // entries equal to T that belong to the answer (the first ones by index)
    len = pow2_at_least(k);
    for (int i = threadIdx.x; i < len; i += kThreads) s_e[i] = 0ull;
    if (threadIdx.x == 0) {
      s_fill = 0;
      s_eq_base = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t i0 = 0; i0 < n; i0 += kThreads) {
      const int64_t i = i0 + threadIdx.x;
      uint32_t key = 0u;
      bool gt = false, eq = false;
      if (i < n) {
        key = ordered_key(row[i * col_stride]);
        gt = key > T;
        eq = key == T;
      }
      // keys above T: all of them, slot order does not matter (sorted afterwards); there are k - need of them
      if (gt) s_e[need + atomicAdd(&s_fill, 1)] = entry(key, (uint32_t)i);
      // keys equal to T: index order decides, so their ranks come from a prefix count over the workgroup
      const unsigned long long m = __ballot(eq);
      if (lane == 0) s_wave[wave][0] = __popcll(m);
      __syncthreads();
      int before = s_eq_base;
      for (int w = 0; w < wave; ++w) before += s_wave[w][0];
      const int rank = before + __popcll(m & ((1ull << lane) - 1ull));
      if (eq && rank < need) s_e[rank] = entry(key, (uint32_t)i);
      __syncthreads();
      if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < kThreads / 64; ++w) total += s_wave[w][0];
        s_eq_base += total;
      }
      __syncthreads();
    }
  }
  sort_desc(s_e, len);
  for (int j = threadIdx.x; j < k; j += kThreads) {
    const unsigned long long e = s_e[j];
    out[j] = (int64_t)(uint32_t)~(uint32_t)(e & 0xffffffffull);
    if (vout) vout[j] = key_value((uint32_t)(e >> 32));
  }
}

}  // namespace

extern "C" int ctr_topk_rows(const float* scores, int64_t row_stride, int64_t col_stride, int64_t rows, int64_t n, int k,
                             int64_t* idx_out, float* val_out, void* stream) {
  CTR_REQUIRE(rows >= 0 && n >= 1 && k >= 1 && k <= n, CTR_EINVAL);
  CTR_REQUIRE(k <= kSortMax && n < (1ll << 31), CTR_ELIMIT);
  if (rows == 0) return CTR_OK;
  CTR_REQUIRE(scores && idx_out, CTR_EINVAL);
  CTR_REQUIRE(rows <= 0x7fffffffll, CTR_ELIMIT);
  hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)rows), dim3(kThreads), 0, (hipStream_t)stream, scores, row_stride,
                     col_stride, n, k, idx_out, val_out);
  return ctr_launch_status();
}
