// Matrix factorisation, fully fused (model/mf.py:23-26): gather two rows, dot,
// sigmoid -- 2 row reads + 16 index bytes in, 4 bytes out per sample.  A group of
// 16 lanes owns one sample (dwordx4 per lane covers E = 64 in one instruction),
// so a wave has 8 independent row pairs in flight.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kGroup = 16;

template <int VEC>
__global__ void __launch_bounds__(kBlock)
mf_fwd_kernel(const float* __restrict__ ut, int64_t nu, const float* __restrict__ it, int64_t ni, int dim,
              const int64_t* __restrict__ uidx, const int64_t* __restrict__ iidx, int64_t batch,
              float* __restrict__ prob, int32_t* err_flag) {
  const int lane = threadIdx.x & (kGroup - 1);
  const int64_t groups = ((int64_t)gridDim.x * blockDim.x) / kGroup;
  for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup; b < batch; b += groups) {
    int64_t u = uidx[b], i = iidx[b];
    if (u < 0 || u >= nu) { if (err_flag) *err_flag = 1; u = 0; }
    if (i < 0 || i >= ni) { if (err_flag) *err_flag = 1; i = 0; }
    const float* up = ut + u * dim;
    const float* ip = it + i * dim;
    float acc = 0.0f;
    if (VEC == 4) {
      for (int e = lane * 4; e < dim; e += kGroup * 4) {
        const float4 a = *reinterpret_cast<const float4*>(up + e);
        const float4 c = *reinterpret_cast<const float4*>(ip + e);
        acc = fmaf(a.x, c.x, acc); acc = fmaf(a.y, c.y, acc); acc = fmaf(a.z, c.z, acc); acc = fmaf(a.w, c.w, acc);
      }
    } else {
      for (int e = lane; e < dim; e += kGroup) acc = fmaf(up[e], ip[e], acc);
    }
    acc = ctr_group_sum<kGroup>(acc);
    if (lane == 0) prob[b] = ctr_sigmoid(acc);
  }
}

// backward: one dword per lane, `group` lanes per sample, so each atomic
// wave-instruction adds to contiguous 4*group-byte runs of gradient rows
__global__ void __launch_bounds__(kBlock)
mf_bwd_kernel(const float* __restrict__ ut, int64_t nu, const float* __restrict__ it, int64_t ni, int dim, int group,
              const int64_t* __restrict__ uidx, const int64_t* __restrict__ iidx, int64_t batch,
              const float* __restrict__ prob, const float* __restrict__ gprob, float* __restrict__ gu,
              float* __restrict__ gi) {
  const int lane = threadIdx.x % group;
  const int64_t groups = ((int64_t)gridDim.x * blockDim.x) / group;
  for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / group; b < batch; b += groups) {
    int64_t u = uidx[b], i = iidx[b];
    if (u < 0 || u >= nu || i < 0 || i >= ni) continue;  // bad id: flagged by the forward, no gradient
    const float p = prob[b];
    const float dz = gprob[b] * p * (1.0f - p);
    const float* up = ut + u * dim;
    const float* ip = it + i * dim;
    for (int e = lane; e < dim; e += group) {
      if (gu) unsafeAtomicAdd(gu + u * dim + e, dz * ip[e]);
      if (gi) unsafeAtomicAdd(gi + i * dim + e, dz * up[e]);
    }
  }
}

int check(const float* ut, int64_t nu, const float* it, int64_t ni, int dim, const int64_t* u, const int64_t* i,
          int64_t batch) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(ut && it && u && i && nu > 0 && ni > 0 && dim > 0, CTR_EINVAL);
  return CTR_OK;
}

}  // namespace

extern "C" int ctr_mf_fwd(const float* user_table, int64_t num_users, const float* item_table, int64_t num_items,
                          int dim, const int64_t* user_idx, const int64_t* item_idx, int64_t batch, float* prob,
                          int32_t* err_flag, void* stream) {
  int rc = check(user_table, num_users, item_table, num_items, dim, user_idx, item_idx, batch);
  if (rc != CTR_OK) return rc;
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(prob, CTR_EINVAL);
  const bool v4 = dim % 4 == 0 && ctr_aligned16(user_table) && ctr_aligned16(item_table);
  const int grid = ctr_stream_grid(batch * kGroup, kBlock);
  hipStream_t st = (hipStream_t)stream;
  if (v4)
    hipLaunchKernelGGL(mf_fwd_kernel<4>, dim3(grid), dim3(kBlock), 0, st, user_table, num_users, item_table,
                       num_items, dim, user_idx, item_idx, batch, prob, err_flag);
  else
    hipLaunchKernelGGL(mf_fwd_kernel<1>, dim3(grid), dim3(kBlock), 0, st, user_table, num_users, item_table,
                       num_items, dim, user_idx, item_idx, batch, prob, err_flag);
  return ctr_launch_status();
}

extern "C" int ctr_mf_bwd(const float* user_table, int64_t num_users, const float* item_table, int64_t num_items,
                          int dim, const int64_t* user_idx, const int64_t* item_idx, int64_t batch,
                          const float* prob, const float* gprob, float* guser, float* gitem, void* stream) {
  int rc = check(user_table, num_users, item_table, num_items, dim, user_idx, item_idx, batch);
  if (rc != CTR_OK) return rc;
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(prob && gprob, CTR_EINVAL);
  int group = 1;
  while (group < dim && group < 64) group <<= 1;
  const int grid = ctr_stream_grid(batch * group, kBlock);
  hipLaunchKernelGGL(mf_bwd_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, user_table, num_users, item_table,
                     num_items, dim, group, user_idx, item_idx, batch, prob, gprob, guser, gitem);
  return ctr_launch_status();
}
