// Two small streaming kernels for models whose first Linear sits on a concatenation of gathered rows, when the tables
// are much smaller than the batch (NeuralCF: Linear(cat(MLP_U[u], MLP_I[i])), reference model/neuralcf.py:43-49):
// the layer is applied to the TABLE ROWS once (a small GEMM), and per sample only
//     out[b, :] = act(A[ia[b], :] + B[ib[b], :])
// remains (ctr_rows_sum_act_fwd); the backward masks the incoming gradient by act'(out) (ctr_act_mask_bwd) and hands
// it to ctr_embed_bwd as the gradient of BOTH gathers.  csrc/ncf_proj.hip is the fused, pinned-shape form of the same
// idea; this is the any-width path.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;

// lane group of (width / 4 rounded up to a power of two, <= 64) lanes per sample, 16 bytes per lane and step
__global__ void __launch_bounds__(kBlock)
rows_sum_act_kernel(const float* __restrict__ ta, const int64_t* __restrict__ ia, int64_t sa, int64_t va,
                    const float* __restrict__ tb, const int64_t* __restrict__ ib, int64_t sb, int64_t vb, int64_t batch,
                    int width, int lpr, int act, float* __restrict__ out, int64_t ldo, int32_t* __restrict__ err_flag) {
  const int per = kBlock / lpr;
  const int sub = threadIdx.x % lpr, which = threadIdx.x / lpr;
  for (int64_t b0 = (int64_t)blockIdx.x * per; b0 < batch; b0 += (int64_t)gridDim.x * per) {
    const int64_t b = b0 + which;
    const int64_t bc = b < batch ? b : batch - 1;
    int64_t ra = ia[bc * sa], rb = ib[bc * sb];
    const bool bad = ra < 0 || ra >= va || rb < 0 || rb >= vb;
    if (bad && b < batch && err_flag) *err_flag = 1;
    ra = (ra < 0 || ra >= va) ? 0 : ra;
    rb = (rb < 0 || rb >= vb) ? 0 : rb;
    for (int c = 4 * sub; c < width; c += 4 * lpr) {
      const ctr_f32x4 x = *reinterpret_cast<const ctr_f32x4*>(ta + ra * width + c);
      const ctr_f32x4 y = *reinterpret_cast<const ctr_f32x4*>(tb + rb * width + c);
      ctr_f32x4 z = x + y;
#pragma unroll
      for (int r = 0; r < 4; ++r) z[r] = ctr_act(z[r], act);
      if (b < batch) *reinterpret_cast<ctr_f32x4*>(out + b * ldo + c) = z;
    }
  }
}

__global__ void __launch_bounds__(kBlock)
act_mask_kernel(float* __restrict__ g, int64_t ldg, const float* __restrict__ y, int64_t ldy, int64_t m, int n4, int act) {
  const int64_t total = m * n4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t row = i / n4;
    const int c = (int)(i - row * n4) * 4;
    ctr_f32x4 gv = *reinterpret_cast<ctr_f32x4*>(g + row * ldg + c);
    const ctr_f32x4 yv = *reinterpret_cast<const ctr_f32x4*>(y + row * ldy + c);
#pragma unroll
    for (int r = 0; r < 4; ++r) gv[r] *= ctr_act_grad(yv[r], act);
    *reinterpret_cast<ctr_f32x4*>(g + row * ldg + c) = gv;
  }
}

}  // namespace

extern "C" int ctr_rows_sum_act_fwd(const float* table_a, const int64_t* idx_a, int64_t stride_a, int64_t vocab_a,
                                    const float* table_b, const int64_t* idx_b, int64_t stride_b, int64_t vocab_b,
                                    int64_t batch, int width, int act, float* out, int64_t ldo, int32_t* err_flag,
                                    void* stream) {
  CTR_REQUIRE(batch >= 0 && width > 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(table_a && table_b && idx_a && idx_b && out && vocab_a > 0 && vocab_b > 0 && ldo >= width, CTR_EINVAL);
  CTR_REQUIRE(act >= CTR_ACT_NONE && act <= CTR_ACT_SIGMOID, CTR_EINVAL);
  CTR_REQUIRE(width % 4 == 0 && ldo % 4 == 0 && ctr_aligned16(table_a) && ctr_aligned16(table_b) && ctr_aligned16(out),
              CTR_EALIGN);
  int lpr = 1;
  while (lpr < width / 4 && lpr < 64) lpr <<= 1;
  const int per = kBlock / lpr;
  hipLaunchKernelGGL(rows_sum_act_kernel, dim3(ctr_stream_grid(batch, per)), dim3(kBlock), 0, (hipStream_t)stream, table_a,
                     idx_a, stride_a, vocab_a, table_b, idx_b, stride_b, vocab_b, batch, width, lpr, act, out, ldo, err_flag);
  return ctr_launch_status();
}

extern "C" int ctr_act_mask_bwd(float* g, int64_t ldg, const float* y, int64_t ldy, int64_t m, int n, int act,
                                void* stream) {
  CTR_REQUIRE(m >= 0 && n > 0, CTR_EINVAL);
  if (m == 0 || act == CTR_ACT_NONE) return CTR_OK;
  CTR_REQUIRE(g && y && ldg >= n && ldy >= n && act <= CTR_ACT_SIGMOID, CTR_EINVAL);
  CTR_REQUIRE(n % 4 == 0 && ldg % 4 == 0 && ldy % 4 == 0 && ctr_aligned16(g) && ctr_aligned16(y), CTR_EALIGN);
  hipLaunchKernelGGL(act_mask_kernel, dim3(ctr_stream_grid(m * (n / 4), kBlock)), dim3(kBlock), 0, (hipStream_t)stream, g,
                     ldg, y, ldy, m, n / 4, act);
  return ctr_launch_status();
}
