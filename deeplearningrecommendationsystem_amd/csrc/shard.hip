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

// ---- the capacity-bounded layout: bucket w owns slots [w*cap, (w+1)*cap) whatever the counts are, so the exchange has
// equal, host-known splits and a FRESH id tensor needs no count exchange and no host read before its ids travel.  One
// pass (bucket starts do not depend on the counts): slot = owner*cap + arrival rank; an id arriving at a full bucket is
// not placed (perm points at the bucket's last slot so that later gathers stay in bounds) and the overflow is reported
// in state[0] -- the caller then falls back to the exact layout.  Unused slots carry -1 on the wire.
__global__ void __launch_bounds__(kBlock)
padded_init_kernel(int32_t* __restrict__ send, int64_t* __restrict__ inv, int64_t slots, int64_t n,
                   int64_t* __restrict__ cursor, int world) {
  if (blockIdx.x == 0 && (int)threadIdx.x <= world) cursor[threadIdx.x] = 0;   // cursor[world]: ids outside the table
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < slots; j += (int64_t)gridDim.x * blockDim.x) {
    send[j] = -1;
    inv[j] = n > 0 ? j % n : 0;   // an unused slot sends SOME row of the caller's gradient: the owner masks it out
  }
}

__global__ void __launch_bounds__(kBlock)
padded_place_kernel(const int64_t* __restrict__ ids, int64_t n, int world, int64_t vocab, int64_t cap,
                    int64_t* __restrict__ cursor, int32_t* __restrict__ send, int64_t* __restrict__ perm,
                    int64_t* __restrict__ inv) {
  __shared__ int s_cnt[kMaxWorld + 1];
  __shared__ int64_t s_base[kMaxWorld];
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += (int64_t)gridDim.x * blockDim.x) {
    if (threadIdx.x <= kMaxWorld) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = i0 + threadIdx.x;
    int owner = 0, rank = 0;
    int64_t local = 0;
    if (i < n) {
      int64_t r = ids[i];
      if (r < 0 || r >= vocab) {
        atomicAdd(&s_cnt[kMaxWorld], 1);
        r = 0;
      }
      owner = (int)(r % world);
      local = r / world;
      rank = atomicAdd(&s_cnt[owner], 1);
    }
    __syncthreads();
    if (threadIdx.x < world && s_cnt[threadIdx.x])
      s_base[threadIdx.x] = (int64_t)atomicAdd(reinterpret_cast<unsigned long long*>(cursor + threadIdx.x),
                                               (unsigned long long)s_cnt[threadIdx.x]);
    if (threadIdx.x == 0 && s_cnt[kMaxWorld])
      atomicAdd(reinterpret_cast<unsigned long long*>(cursor + world), (unsigned long long)s_cnt[kMaxWorld]);
    __syncthreads();
    if (i < n) {
      const int64_t at = s_base[owner] + rank;
      const int64_t slot = (int64_t)owner * cap + (at < cap ? at : cap - 1);
      perm[i] = slot;
      if (at < cap) {
        send[slot] = (int32_t)local;
        inv[slot] = i;
      }
    }
    __syncthreads();
  }
}

// state = {a bucket overflowed, ids outside the table, n, -n}: one MAX all-reduce tells every rank whether ANY rank must
// fall back / raise and whether the ranks passed different id counts (max n != -max(-n))
__global__ void padded_state_kernel(const int64_t* __restrict__ cursor, int world, int64_t cap, int64_t n,
                                    int64_t* __restrict__ state) {
  if (threadIdx.x != 0) return;
  int64_t over = 0;
  for (int w = 0; w < world; ++w) over |= cursor[w] > cap ? 1 : 0;
  state[0] = over;
  state[1] = cursor[world];
  state[2] = n;
  state[3] = -n;
}

// what the owner makes of the received slots: rows to gather / scatter (an unused slot reads and "updates" -- by an
// all-zero row -- a row of its own, slot % local_rows, so that no row becomes a hot spot of atomics), the 0/1 mask of
// the real slots, and the ids as the sparse-mode row marker wants them (-1 = not a row)
__global__ void __launch_bounds__(kBlock)
recv_rows_kernel(const int32_t* __restrict__ recv, int64_t slots, int64_t local_rows, int64_t* __restrict__ rows,
                 float* __restrict__ valid, int64_t* __restrict__ mark) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < slots; j += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = recv[j];
    const bool ok = r >= 0 && r < local_rows;
    rows[j] = ok ? (int64_t)r : j % local_rows;
    valid[j] = ok ? 1.0f : 0.0f;
    mark[j] = ok ? (int64_t)r : -1;
  }
}

// table[idx[i]][:] = 0 (dwordx4 lanes when the rows allow it): clears the rows a step touched in a persistent
// shard-gradient buffer
template <int VEC>
__global__ void __launch_bounds__(kBlock)
rows_zero_kernel(float* __restrict__ table, int64_t ld, int dim, const int64_t* __restrict__ idx, int64_t n,
                 int64_t rows) {
  const int per_row = dim / VEC;
  const int64_t total = n * per_row;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx[e / per_row];
    if (r < 0 || r >= rows) continue;
    float* dst = table + r * ld + (e % per_row) * VEC;
    if (VEC == 4) {
      *reinterpret_cast<ctr_f32x4*>(dst) = ctr_f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      dst[0] = 0.0f;
    }
  }
}

}  // namespace

extern "C" int ctr_shard_bucket_padded(const int64_t* ids, int64_t n, int world, int64_t vocab, int64_t cap,
                                       int64_t* cursor, int32_t* send, int64_t* perm, int64_t* inv, int64_t* state,
                                       void* stream) {
  CTR_REQUIRE(n >= 0 && world >= 1 && world <= kMaxWorld && cap >= 1 && cursor && send && inv && state && vocab > 0,
              CTR_EINVAL);
  CTR_REQUIRE(vocab / world < (1ll << 31), CTR_ELIMIT);
  hipStream_t st = (hipStream_t)stream;
  const int64_t slots = (int64_t)world * cap;
  hipLaunchKernelGGL(padded_init_kernel, dim3(ctr_stream_grid(slots, kBlock)), dim3(kBlock), 0, st, send, inv, slots, n,
                     cursor, world);
  if (n > 0) {
    CTR_REQUIRE(ids && perm, CTR_EINVAL);
    hipLaunchKernelGGL(padded_place_kernel, dim3(ctr_stream_grid(n, kBlock)), dim3(kBlock), 0, st, ids, n, world, vocab,
                       cap, cursor, send, perm, inv);
  }
  hipLaunchKernelGGL(padded_state_kernel, dim3(1), dim3(64), 0, st, cursor, world, cap, n, state);
  return ctr_launch_status();
}

extern "C" int ctr_shard_recv_rows(const int32_t* recv, int64_t slots, int64_t local_rows, int64_t* rows, float* valid,
                                   int64_t* mark, void* stream) {
  CTR_REQUIRE(slots >= 0 && local_rows >= 1, CTR_EINVAL);
  if (slots == 0) return CTR_OK;
  CTR_REQUIRE(recv && rows && valid && mark, CTR_EINVAL);
  hipLaunchKernelGGL(recv_rows_kernel, dim3(ctr_stream_grid(slots, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, recv,
                     slots, local_rows, rows, valid, mark);
  return ctr_launch_status();
}

extern "C" int ctr_rows_zero(float* table, int64_t ld, int64_t rows, int dim, const int64_t* idx, int64_t n,
                             void* stream) {
  CTR_REQUIRE(n >= 0 && rows >= 0 && dim > 0 && ld >= dim, CTR_EINVAL);
  if (n == 0 || rows == 0) return CTR_OK;
  CTR_REQUIRE(table && idx, CTR_EINVAL);
  hipStream_t st = (hipStream_t)stream;
  if (dim % 4 == 0 && ld % 4 == 0 && ctr_aligned16(table))
    hipLaunchKernelGGL((rows_zero_kernel<4>), dim3(ctr_stream_grid(n * (dim / 4), kBlock)), dim3(kBlock), 0, st, table, ld,
                       dim, idx, n, rows);
  else
    hipLaunchKernelGGL((rows_zero_kernel<1>), dim3(ctr_stream_grid(n * dim, kBlock)), dim3(kBlock), 0, st, table, ld, dim,
                       idx, n, rows);
  return ctr_launch_status();
}

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
