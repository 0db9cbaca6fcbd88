// A linear layer W (n x k, bias b) whose output feeds ONLY a single-unit layer u is the same map as
// a k-wide dot product:   (h W^T + b) . u + b2  ==  h . v + c,   v = W^T u,  c = b . u + b2.
// NeuralCF ends like that (model/neuralcf.py:27,50-56: linear 8 -> mf_dim, cat with the GMF vector,
// linear2 2*mf_dim -> 1): folding the pair per step removes the 8 -> 64 layer, its (B, 64) output and
// that output's gradient from the batch-sized work; what is left here is O(n k) per step.
//   u_full = [ u_pass (p columns that stay as they are: the GMF half) | u (n columns fed by W) ]
// forward : wfold = [ u_pass | W^T u ] (p + k),  cfold = b . u + b2
// backward: from gwfold = d/d wfold (p + k) and gc = d/d cfold:
//           gu_full += [ gwfold[:p] | W gwfold[p:] + b gc ],  gW += u (x) gwfold[p:],  gb += u gc,  gb2 += gc
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;

// One workgroup.  Column j of W^T u is summed by S = 256 / k threads over S slices of the rows (all loads
// in flight at once: a thread per column walking the n rows took 10 us of dependent loads), then the S
// partials are added in slice order -- the result does not depend on scheduling.
__global__ void __launch_bounds__(kBlock)
fold_head_fwd_kernel(const float* __restrict__ u_full, int p, const float* __restrict__ w, int64_t ldw,
                     const float* __restrict__ b, const float* __restrict__ b2, int n, int k, float* __restrict__ wfold,
                     float* __restrict__ cfold) {
  __shared__ float s_part[kBlock];
  __shared__ float s_c[kBlock / 64];
  const float* u = u_full + p;
  for (int j = threadIdx.x; j < p; j += kBlock) wfold[j] = u_full[j];
  // c = b . u + b2
  {
    float t = 0.0f;
    if (b)
      for (int i = threadIdx.x; i < n; i += kBlock) t = fmaf(b[i], u[i], t);
    t = ctr_wave_sum(t);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = t;
  }
  for (int j0 = 0; j0 < k; j0 += kBlock) {
    const int kc = k - j0 < kBlock ? k - j0 : kBlock;  // columns of this pass
    const int slices = kBlock / kc;                     // >= 1
    const int j = threadIdx.x % kc, sl = threadIdx.x / kc;
    float t = 0.0f;
    if (sl < slices) {
      const int rows = (n + slices - 1) / slices;
      const int i1 = (sl + 1) * rows < n ? (sl + 1) * rows : n;
      for (int i = sl * rows; i < i1; ++i) t = fmaf(w[(int64_t)i * ldw + j0 + j], u[i], t);
    }
    __syncthreads();
    s_part[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x < kc) {
      float v = 0.0f;
      for (int q = 0; q < slices; ++q) v += s_part[q * kc + threadIdx.x];
      wfold[p + j0 + threadIdx.x] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = b2 ? b2[0] : 0.0f;
    for (int wv = 0; wv < kBlock / 64; ++wv) t += s_c[wv];
    cfold[0] = t;
  }
}

__global__ void __launch_bounds__(kBlock)
fold_head_bwd_kernel(const float* __restrict__ u_full, int p, const float* __restrict__ w, int64_t ldw,
                     const float* __restrict__ b, int n, int k, const float* __restrict__ gwfold,
                     const float* __restrict__ gc, float* __restrict__ gu_full, float* __restrict__ gw, int64_t ldgw,
                     float* __restrict__ gb, float* __restrict__ gb2) {
  const float* u = u_full + p;
  const float* gv = gwfold + p;
  const float c = gc[0];
  const int total = p + n + n * k + n + 1;
  for (int t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
    int j = t;
    if (j < p) {
      if (gu_full) gu_full[j] += gwfold[j];
      continue;
    }
    j -= p;
    if (j < n) {
      if (gu_full) {
        float s = b ? b[j] * c : 0.0f;
        for (int q = 0; q < k; ++q) s = fmaf(w[(int64_t)j * ldw + q], gv[q], s);
        gu_full[p + j] += s;
      }
      continue;
    }
    j -= n;
    if (j < n * k) {
      const int i = j / k, q = j - i * k;
      if (gw) gw[(int64_t)i * ldgw + q] += u[i] * gv[q];
      continue;
    }
    j -= n * k;
    if (j < n) {
      if (gb) gb[j] += u[j] * c;
      continue;
    }
    if (gb2) gb2[0] += c;
  }
}

}  // namespace

extern "C" int ctr_fold_head_fwd(const float* u_full, int p, const float* w, int64_t ldw, const float* b, const float* b2,
                                 int n, int k, float* wfold, float* cfold, void* stream) {
  CTR_REQUIRE(u_full && w && wfold && cfold && p >= 0 && n > 0 && k > 0 && ldw >= k, CTR_EINVAL);
  hipLaunchKernelGGL(fold_head_fwd_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, u_full, p, w, ldw, b, b2, n, k,
                     wfold, cfold);
  return ctr_launch_status();
}

extern "C" int ctr_fold_head_bwd(const float* u_full, int p, const float* w, int64_t ldw, const float* b, int n, int k,
                                 const float* gwfold, const float* gc, float* gu_full, float* gw, int64_t ldgw, float* gb,
                                 float* gb2, void* stream) {
  CTR_REQUIRE(u_full && w && gwfold && gc && p >= 0 && n > 0 && k > 0 && ldw >= k && (!gw || ldgw >= k), CTR_EINVAL);
  const int64_t total = (int64_t)p + n + (int64_t)n * k + n + 1;
  hipLaunchKernelGGL(fold_head_bwd_kernel, dim3(ctr_stream_grid(total, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     u_full, p, w, ldw, b, n, k, gwfold, gc, gu_full, gw, ldgw, gb, gb2);
  return ctr_launch_status();
}
