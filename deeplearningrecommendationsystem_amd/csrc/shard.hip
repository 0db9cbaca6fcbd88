// Row-sharded embedding tables (one process per GPU, owner(row) = row % world,
// local row = row / world): bucket a batch of global ids by owning rank before the
// RCCL all-to-all.  Two passes over the ids (HBM-bound, 8 B read + 16 B written per id):
//   count : per-workgroup LDS histogram of owners -> counts[world]; ids outside [0, vocab) are counted in
//           counts[world] (the host reads the counts anyway to size the exchange and raises IndexError) and
//           travel as row 0 of rank 0
//   place : slot = start[owner] + cursor[owner]++  (the order inside a bucket is not fixed;
//           `perm` / `inv` record it so the rows can be put back in batch order)
// outputs: send[slot] = local row as int32 (what goes on the wire: half the bytes of the int64 ids),
// perm[i] = slot, inv[slot] = i.  counts / cursor are zeroed by a kernel of their own: no hipMemsetAsync in
// the library (hipGraph note), so the bucketing can sit inside a captured step.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxWorld = 64;

__global__ void zero_counters_kernel(int64_t* __restrict__ counts, int64_t* __restrict__ cursor, int world) {
  if ((int)threadIdx.x <= world) counts[threadIdx.x] = 0;   // counts has world + 1 slots (the last: bad ids)
  if ((int)threadIdx.x < world) cursor[threadIdx.x] = 0;
}

__global__ void __launch_bounds__(kBlock)
owner_count_kernel(const int64_t* __restrict__ ids, int64_t n, int world, int64_t vocab, int64_t* __restrict__ counts) {
  __shared__ int s_cnt[kMaxWorld + 1];
  if (threadIdx.x <= kMaxWorld) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = ids[i];
    if (r < 0 || r >= vocab) {
      atomicAdd(&s_cnt[kMaxWorld], 1);
      r = 0;
    }
    atomicAdd(&s_cnt[(int)(r % world)], 1);
  }
  __syncthreads();
  if (threadIdx.x < world && s_cnt[threadIdx.x])
    atomicAdd(reinterpret_cast<unsigned long long*>(counts + threadIdx.x), (unsigned long long)s_cnt[threadIdx.x]);
  if (threadIdx.x == 0 && s_cnt[kMaxWorld])
    atomicAdd(reinterpret_cast<unsigned long long*>(counts + world), (unsigned long long)s_cnt[kMaxWorld]);
}

__global__ void __launch_bounds__(kBlock)
owner_place_kernel(const int64_t* __restrict__ ids, int64_t n, int world, int64_t vocab,
                   const int64_t* __restrict__ counts, int64_t* __restrict__ cursor, int32_t* __restrict__ send,
                   int64_t* __restrict__ perm, int64_t* __restrict__ inv) {
  __shared__ int64_t s_start[kMaxWorld];
  __shared__ int s_cnt[kMaxWorld];
  __shared__ int64_t s_base[kMaxWorld];
  if (threadIdx.x == 0) {
    int64_t acc = 0;
    for (int w = 0; w < world; ++w) {
      s_start[w] = acc;
      acc += counts[w];
    }
  }
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += (int64_t)gridDim.x * blockDim.x) {
    if (threadIdx.x < kMaxWorld) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    // one global cursor bump per (workgroup pass, owner), ranks inside it from LDS
    const int64_t i = i0 + threadIdx.x;
    int owner = 0, rank = 0;
    int64_t local = 0;
    if (i < n) {
      int64_t r = ids[i];
      if (r < 0 || r >= vocab) r = 0;
      owner = (int)(r % world);
      local = r / world;
      rank = atomicAdd(&s_cnt[owner], 1);
    }
    __syncthreads();
    if (threadIdx.x < world && s_cnt[threadIdx.x])
      s_base[threadIdx.x] = (int64_t)atomicAdd(reinterpret_cast<unsigned long long*>(cursor + threadIdx.x),
                                               (unsigned long long)s_cnt[threadIdx.x]);
    __syncthreads();
    if (i < n) {
      const int64_t slot = s_start[owner] + s_base[owner] + rank;
      send[slot] = (int32_t)local;
      perm[i] = slot;
      inv[slot] = i;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int ctr_shard_bucket(const int64_t* ids, int64_t n, int world, int64_t vocab, int64_t* counts,
                                int64_t* cursor, int32_t* send, int64_t* perm, int64_t* inv, void* stream) {
  CTR_REQUIRE(n >= 0 && world >= 1 && world <= kMaxWorld && counts && cursor && vocab > 0, CTR_EINVAL);
  CTR_REQUIRE(vocab / world < (1ll << 31), CTR_ELIMIT);  // local rows travel as int32
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(zero_counters_kernel, dim3(1), dim3(kMaxWorld + 64), 0, st, counts, cursor, world);
  if (n == 0) return ctr_launch_status();
  CTR_REQUIRE(ids && send && perm && inv, CTR_EINVAL);
  const int grid = ctr_stream_grid(n, kBlock);
  hipLaunchKernelGGL(owner_count_kernel, dim3(grid), dim3(kBlock), 0, st, ids, n, world, vocab, counts);
  hipLaunchKernelGGL(owner_place_kernel, dim3(grid), dim3(kBlock), 0, st, ids, n, world, vocab, counts, cursor, send, perm,
                     inv);
  return ctr_launch_status();
}
