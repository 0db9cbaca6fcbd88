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

// ---------------------------------------------------------------------------------------------------
// Gradient of two (V, 1) first-order tables indexed by the user / item id columns of the feature matrix
// (model/ffm.py:19-26 `user` / `item`, deepfm.py / widedeep.py / lr.py `*_1st`): guser1[u_b] += v_b, gitem1[i_b] += v_b,
// v_b = g[b] (* p_b (1 - p_b) when `prob` is given: the sigmoid head's dlogit).
// Done inside the interaction kernels this was one 4-byte atomic per sample and table on 943 / 1682 packed floats: 59 /
// 105 cache lines, ~1100 / 620 adds each, and same-LINE atomics are served one after the other at the memory side
// (~30 ns): a 33 us chain under ffm_fused_bwd (48 us) and fm_wide_bwd (36 us of logistic regression's 66 us step).
// Here a workgroup sums its slice of the batch in LDS (ds_add_f32 on nu + ni floats) and flushes with one dword per
// lane on consecutive rows, which the memory side takes as one 64-byte operation per line and workgroup.
namespace {

__global__ void __launch_bounds__(256)
rows1_scatter_kernel(const float* __restrict__ x, int64_t ldx, int user_col, int item_col, const float* __restrict__ g,
                     int64_t ldg, const float* __restrict__ prob, int64_t ldp, int64_t batch, float* __restrict__ guser1,
                     int64_t nu, float* __restrict__ gitem1, int64_t ni) {
  extern __shared__ float s_acc[];   // [nu | ni]
  for (int64_t v = threadIdx.x; v < nu + ni; v += blockDim.x) s_acc[v] = 0.0f;
  __syncthreads();
  const int64_t per = (batch + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < batch ? lo + per : batch;
  constexpr int kU = 4;
  for (int64_t b0 = lo + threadIdx.x; b0 < hi; b0 += (int64_t)blockDim.x * kU) {
    float uf[kU], itf[kU], gv[kU], pv[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      const int64_t b = b0 + (int64_t)k * blockDim.x, bc = b < hi ? b : hi - 1;   // unconditional loads, masked below
      uf[k] = x[bc * ldx + user_col];
      itf[k] = x[bc * ldx + item_col];
      gv[k] = g[bc * ldg];
      pv[k] = prob ? prob[bc * ldp] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      const int64_t b = b0 + (int64_t)k * blockDim.x;
      if (b < hi) {
        const float v = prob ? gv[k] * pv[k] * (1.0f - pv[k]) : gv[k];
        const int64_t u = (int64_t)uf[k], it = (int64_t)itf[k];
        if (guser1 && u >= 0 && u < nu) atomicAdd(&s_acc[u], v);
        if (gitem1 && it >= 0 && it < ni) atomicAdd(&s_acc[nu + it], v);
      }
    }
  }
  __syncthreads();
  if (guser1)
    for (int64_t v = threadIdx.x; v < nu; v += blockDim.x) {
      const float t = s_acc[v];
      if (t != 0.0f) ctr_atomic_add_global(guser1 + v, t);
    }
  if (gitem1)
    for (int64_t v = threadIdx.x; v < ni; v += blockDim.x) {
      const float t = s_acc[nu + v];
      if (t != 0.0f) ctr_atomic_add_global(gitem1 + v, t);
    }
}

}  // namespace

extern "C" int ctr_rows1_scatter(const float* x, int64_t ldx, int user_col, int item_col, const float* g, int64_t ldg,
                                 const float* prob, int64_t ldp, int64_t batch, float* guser1, int64_t num_users,
                                 float* gitem1, int64_t num_items, void* stream) {
  CTR_REQUIRE(batch >= 0 && num_users > 0 && num_items > 0 && user_col >= 0 && item_col >= 0, CTR_EINVAL);
  if (batch == 0 || (!guser1 && !gitem1)) return CTR_OK;
  CTR_REQUIRE(x && g && ldx > user_col && ldx > item_col && ldg >= 1 && (!prob || ldp >= 1), CTR_EINVAL);
  CTR_REQUIRE(num_users + num_items <= CTR_ROWS1_MAX_ROWS, CTR_ELIMIT);
  // few workgroups: each ends with one flush per touched line (same-line operations queue up at the memory side)
  int64_t grid = ctr_ceil_div(batch, 1024);
  if (grid > 64) grid = 64;
  const size_t lds = (size_t)(num_users + num_items) * sizeof(float);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(rows1_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return CTR_ELAUNCH;
  hipLaunchKernelGGL(rows1_scatter_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, x, ldx, user_col,
                     item_col, g, ldg, prob, ldp, batch, guser1, num_users, gitem1, num_items);
  return ctr_launch_status();
}
