// Cross layer of Deep & Cross (reference model/deepcross.py:7-18):
//     x_{l+1} = x0 * (W_l x_l) + b_l + x_l          (W_l: bias-free d x d Linear)
// The d x d product is a ctr_linear_fwd call (matrix cores); these kernels are the
// elementwise combine around it and its backward.  HBM-bound streaming: one pass, dwordx4
// where rows are 16-byte aligned, grid-stride over the (row, column-vector) space.
//   fwd:  y = x0 * u + bias + xl
//   bwd:  gu  = gy * x0            (operand of the layer's dX / dW GEMMs)
//         gx0 += gy * u            (accumulated over the layers)
//         gb  = column sums of gy  (per-workgroup partials in the workspace, fixed-order sum)
//   the gradient w.r.t. x_l is gy itself plus gu W_l: the caller accumulates the GEMM's dX
//   straight into the gy buffer.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;

struct CrossArgs {
  const float* x0; int64_t ld0;
  const float* u; int64_t ldu;
  const float* xl; int64_t ldl;
  const float* bias;
  float* y; int64_t ldy;
  int64_t m; int d;
};

template <int VEC>
__global__ void __launch_bounds__(kBlock) cross_fwd_kernel(const CrossArgs a) {
  const int cpr = (a.d + VEC - 1) / VEC;  // column vectors per row
  const int64_t total = a.m * cpr;
  for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < total; g += (int64_t)gridDim.x * kBlock) {
    const int64_t r = g / cpr;
    const int c = (int)(g - r * cpr) * VEC;
    if (VEC == 4 && c + 3 < a.d) {
      const float4 p = ctr_ldg(reinterpret_cast<const float4*>(a.x0 + r * a.ld0 + c));
      const float4 q = ctr_ldg(reinterpret_cast<const float4*>(a.u + r * a.ldu + c));
      const float4 s = ctr_ldg(reinterpret_cast<const float4*>(a.xl + r * a.ldl + c));
      float4 o;
      o.x = fmaf(p.x, q.x, a.bias[c + 0]) + s.x;
      o.y = fmaf(p.y, q.y, a.bias[c + 1]) + s.y;
      o.z = fmaf(p.z, q.z, a.bias[c + 2]) + s.z;
      o.w = fmaf(p.w, q.w, a.bias[c + 3]) + s.w;
      ctr_stg(reinterpret_cast<float4*>(a.y + r * a.ldy + c), o);
    } else {
      for (int v = 0; v < VEC && c + v < a.d; ++v)
        a.y[r * a.ldy + c + v] = fmaf(a.x0[r * a.ld0 + c + v], a.u[r * a.ldu + c + v], a.bias[c + v]) + a.xl[r * a.ldl + c + v];
    }
  }
}

struct CrossBwdArgs {
  const float* x0; int64_t ld0;
  const float* u; int64_t ldu;
  const float* gy; int64_t ldg;
  float* gu; int64_t ldgu;
  float* gx0; int64_t ldgx0;
  float* ws;  // [gridDim.x][d] partial column sums of gy
  int64_t m; int d;
};

// a workgroup owns a contiguous block of rows; thread t owns columns t, t + 256, ... so the
// column sums stay in registers (d <= 4 * 256)
__global__ void __launch_bounds__(kBlock) cross_bwd_kernel(const CrossBwdArgs a, int64_t rows_per_block) {
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < a.m ? r0 + rows_per_block : a.m;
  float sum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t r = r0; r < r1; ++r) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = threadIdx.x + q * kBlock;
      if (c < a.d) {
        const float g = a.gy[r * a.ldg + c];
        a.gu[r * a.ldgu + c] = g * a.x0[r * a.ld0 + c];
        a.gx0[r * a.ldgx0 + c] += g * a.u[r * a.ldu + c];
        sum[q] += g;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = threadIdx.x + q * kBlock;
    if (c < a.d) a.ws[(int64_t)blockIdx.x * a.d + c] = sum[q];
  }
}

}  // namespace

extern "C" int ctr_cross_fwd(const float* x0, int64_t ldx0, const float* u, int64_t ldu, const float* xl, int64_t ldxl,
                             const float* bias, float* y, int64_t ldy, int64_t m, int d, void* stream) {
  CTR_REQUIRE(m >= 0 && d > 0, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  CTR_REQUIRE(x0 && u && xl && bias && y && ldx0 >= d && ldu >= d && ldxl >= d && ldy >= d, CTR_EINVAL);
  const bool vec = ctr_aligned16(x0) && ctr_aligned16(u) && ctr_aligned16(xl) && ctr_aligned16(y) && ldx0 % 4 == 0 &&
                   ldu % 4 == 0 && ldxl % 4 == 0 && ldy % 4 == 0;
  const CrossArgs a{x0, ldx0, u, ldu, xl, ldxl, bias, y, ldy, m, d};
  const int grid = ctr_stream_grid(m * ((d + 3) / 4), kBlock);
  if (vec)
    hipLaunchKernelGGL(cross_fwd_kernel<4>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(cross_fwd_kernel<1>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a);
  return ctr_launch_status();
}

extern "C" int ctr_cross_bwd(const float* x0, int64_t ldx0, const float* u, int64_t ldu, const float* gy, int64_t ldgy,
                             float* gu, int64_t ldgu, float* gx0, int64_t ldgx0, float* gbias, int64_t m, int d,
                             float* workspace, int64_t workspace_floats, void* stream) {
  CTR_REQUIRE(m >= 0 && d > 0, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  CTR_REQUIRE(x0 && u && gy && gu && gx0 && gbias && workspace, CTR_EINVAL);
  CTR_REQUIRE(ldx0 >= d && ldu >= d && ldgy >= d && ldgu >= d && ldgx0 >= d, CTR_EINVAL);
  CTR_REQUIRE(d <= 4 * kBlock, CTR_ELIMIT);
  int64_t blocks = ctr_ceil_div(m, 32);
  if (blocks > 2048) blocks = 2048;
  if (blocks * d > workspace_floats) blocks = workspace_floats / d;
  CTR_REQUIRE(blocks >= 1, CTR_ELIMIT);
  const int64_t rows_per_block = ctr_ceil_div(m, blocks);
  blocks = ctr_ceil_div(m, rows_per_block);
  const CrossBwdArgs a{x0, ldx0, u, ldu, gy, ldgy, gu, ldgu, gx0, ldgx0, workspace, m, d};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cross_bwd_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, a, rows_per_block);
  int rc = ctr_launch_status();
  if (rc != CTR_OK) return rc;
  CtrSegments segs;
  segs.n = 1;
  segs.s[0] = CtrSegment{0, d, gbias};
  return ctr_reduce_segments(workspace, (int)blocks, d, segs, st);
}
