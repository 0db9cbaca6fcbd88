// Row-sharded embedding tables (one process per GPU, owner(row) = row % world,
// local row = row / world): bucket a batch of global ids by owning rank before the
// RCCL all-to-all.  Two passes over the ids (HBM-bound, 8 B read + 16 B written per id), each workgroup on its own
// contiguous slice of the batch:
//   count : LDS histogram of the slice's owners -> hist[workgroup][0..world] in the scratch; ids outside [0, vocab) are
//           counted in [world] (the host reads the counts anyway to size the exchange and raises IndexError) and travel
//           as row 0 of rank 0
//   place : slot = (ids of lower owners) + (this owner's ids in the slices in front) + rank inside the slice, from the
//           histograms and an LDS counter -- no global atomic (the order inside a bucket is the batch order of the
//           slices, inside a 256-id pass it is not fixed; `perm` / `inv` record it)
// outputs: send[slot] = local row as int32 (what goes on the wire: half the bytes of the int64 ids),
// perm[i] = slot, inv[slot] = i.  No hipMemsetAsync in the library (hipGraph note): every scratch word a
// launch reads is written by the launch before it, so the bucketing can sit inside a captured step.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxWorld = 64;

// Both passes give workgroup w the contiguous slice [w*chunk, (w+1)*chunk) of the ids, so the placement needs no global
// atomic: `count` stores the slice's histogram hist[w][0..world] (plain stores; [world] = ids outside the table), `place`
// sums the histograms of the workgroups in front of it.  (Round 2 bumped one global cursor per owner and workgroup pass:
// n / 256 returning atomics per owner on ONE 64-byte line, served one after the other at ~30 ns -- 0.4 ms for the 3.3 M
// ids of a DIN batch, most of what a fresh id tensor's exchange plan cost.)
constexpr int kMaxGrid = 256;

__global__ void __launch_bounds__(kBlock)
owner_count_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t chunk, int world, int64_t vocab,
                   int64_t* __restrict__ hist) {
  __shared__ int s_cnt[kMaxWorld + 1];
  if (threadIdx.x <= kMaxWorld) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int64_t r = ids[i];
    if (r < 0 || r >= vocab) {
      atomicAdd(&s_cnt[kMaxWorld], 1);
      r = 0;
    }
    atomicAdd(&s_cnt[(int)(r % world)], 1);
  }
  __syncthreads();
  if ((int)threadIdx.x <= world)
    hist[(int64_t)blockIdx.x * (world + 1) + threadIdx.x] = s_cnt[(int)threadIdx.x < world ? threadIdx.x : kMaxWorld];
}

// s_first[o] = first slot of this workgroup's ids of owner o; totals[o] = ids of owner o in the whole batch
__device__ __forceinline__ void slice_bases(const int64_t* __restrict__ hist, int world, int64_t* s_first, int64_t* s_total,
                                            int64_t* s_before) {
  // thread (o, part): owner o, workgroups part, part + parts, ... ; then a tree over the parts in LDS
  const int o = threadIdx.x % (kMaxWorld + 1), part = threadIdx.x / (kMaxWorld + 1), parts = kBlock / (kMaxWorld + 1);
  __shared__ int64_t s_t[kBlock], s_b[kBlock];
  int64_t tot = 0, bef = 0;
  if (o <= world && part < parts)
    for (int w = part; w < (int)gridDim.x; w += parts) {
      const int64_t c = hist[(int64_t)w * (world + 1) + o];
      tot += c;
      if (w < (int)blockIdx.x) bef += c;
    }
  s_t[threadIdx.x] = tot;
  s_b[threadIdx.x] = bef;
  __syncthreads();
  if (part == 0 && o <= world) {
    for (int p2 = 1; p2 < parts; ++p2) {
      tot += s_t[p2 * (kMaxWorld + 1) + o];
      bef += s_b[p2 * (kMaxWorld + 1) + o];
    }
    s_total[o] = tot;
    s_before[o] = bef;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t acc = 0;
    for (int w = 0; w < world; ++w) {
      s_first[w] = acc + s_before[w];
      acc += s_total[w];
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(kBlock)
owner_place_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t chunk, int world, int64_t vocab,
                   const int64_t* __restrict__ hist, int64_t* __restrict__ counts, int32_t* __restrict__ send,
                   int64_t* __restrict__ perm, int64_t* __restrict__ inv) {
  __shared__ int64_t s_first[kMaxWorld], s_total[kMaxWorld + 1], s_before[kMaxWorld + 1];
  __shared__ int s_cnt[kMaxWorld];
  slice_bases(hist, world, s_first, s_total, s_before);
  if (blockIdx.x == 0 && (int)threadIdx.x <= world) counts[threadIdx.x] = s_total[threadIdx.x];
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for (int64_t i0 = lo; i0 < hi; i0 += blockDim.x) {
    if (threadIdx.x < kMaxWorld) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = i0 + threadIdx.x;
    int owner = 0, rank = 0;
    int64_t local = 0;
    if (i < hi) {
      int64_t r = ids[i];
      if (r < 0 || r >= vocab) r = 0;
      owner = (int)(r % world);
      local = r / world;
      rank = atomicAdd(&s_cnt[owner], 1);
    }
    __syncthreads();
    if (i < hi) {
      const int64_t slot = s_first[owner] + rank;
      send[slot] = (int32_t)local;
      perm[i] = slot;
      inv[slot] = i;
    }
    __syncthreads();
    if ((int)threadIdx.x < world) s_first[threadIdx.x] += s_cnt[threadIdx.x];
    __syncthreads();
  }
}

// ---- the capacity-bounded layout: bucket w owns slots [w*cap, (w+1)*cap) whatever the counts are, so the exchange has
// equal, host-known splits and a FRESH id tensor needs no count exchange and no host read before its ids travel.  One
// pass (bucket starts do not depend on the counts): slot = owner*cap + arrival rank; an id arriving at a full bucket is
// not placed (perm points at the bucket's last slot so that later gathers stay in bounds) and the overflow is reported
// in state[0] -- the caller then falls back to the exact layout.  Unused slots carry -1 on the wire.
__global__ void __launch_bounds__(kBlock)
padded_init_kernel(int32_t* __restrict__ send, int64_t* __restrict__ inv, int64_t slots, int64_t n) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < slots; j += (int64_t)gridDim.x * blockDim.x) {
    send[j] = -1;
    inv[j] = n > 0 ? j % n : 0;   // an unused slot sends SOME row of the caller's gradient: the owner masks it out
  }
}

// same slices and histograms as the exact layout (owner_count_kernel): slot = owner*cap + ids of that owner in front
__global__ void __launch_bounds__(kBlock)
padded_place_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t chunk, int world, int64_t vocab, int64_t cap,
                    const int64_t* __restrict__ hist, int32_t* __restrict__ send, int64_t* __restrict__ perm,
                    int64_t* __restrict__ inv, int64_t* __restrict__ state) {
  __shared__ int64_t s_first[kMaxWorld], s_total[kMaxWorld + 1], s_before[kMaxWorld + 1];
  __shared__ int s_cnt[kMaxWorld];
  slice_bases(hist, world, s_first, s_total, s_before);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // state = {a bucket overflowed, ids outside the table, n, -n}: one MAX all-reduce tells every rank whether ANY rank
    // must fall back / raise and whether the ranks passed different id counts (max n != -max(-n))
    int64_t over = 0;
    for (int w = 0; w < world; ++w) over |= s_total[w] > cap ? 1 : 0;
    state[0] = over;
    state[1] = s_total[world];
    state[2] = n;
    state[3] = -n;
  }
  if ((int)threadIdx.x < world) s_first[threadIdx.x] = s_before[threadIdx.x];   // position inside the owner's bucket
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for (int64_t i0 = lo; i0 < hi; i0 += blockDim.x) {
    if (threadIdx.x < kMaxWorld) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = i0 + threadIdx.x;
    int owner = 0, rank = 0;
    int64_t local = 0;
    if (i < hi) {
      int64_t r = ids[i];
      if (r < 0 || r >= vocab) r = 0;
      owner = (int)(r % world);
      local = r / world;
      rank = atomicAdd(&s_cnt[owner], 1);
    }
    __syncthreads();
    if (i < hi) {
      const int64_t at = s_first[owner] + rank;
      const int64_t slot = (int64_t)owner * cap + (at < cap ? at : cap - 1);
      perm[i] = slot;
      if (at < cap) {
        send[slot] = (int32_t)local;
        inv[slot] = i;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < world) s_first[threadIdx.x] += s_cnt[threadIdx.x];
    __syncthreads();
  }
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

static inline void slices(int64_t n, int* grid, int64_t* chunk) {
  int64_t g = ctr_ceil_div(n, 8 * kBlock);           // at least eight passes of a workgroup per slice
  if (g > kMaxGrid) g = kMaxGrid;
  if (g < 1) g = 1;
  *grid = (int)g;
  *chunk = ctr_ceil_div(n, g);
}

extern "C" int ctr_shard_bucket_padded(const int64_t* ids, int64_t n, int world, int64_t vocab, int64_t cap,
                                       int64_t* cursor, int32_t* send, int64_t* perm, int64_t* inv, int64_t* state,
                                       void* stream) {
  CTR_REQUIRE(n >= 0 && world >= 1 && world <= kMaxWorld && cap >= 1 && cursor && send && inv && state && vocab > 0,
              CTR_EINVAL);
  CTR_REQUIRE(vocab / world < (1ll << 31), CTR_ELIMIT);
  hipStream_t st = (hipStream_t)stream;
  const int64_t slots = (int64_t)world * cap;
  int grid;
  int64_t chunk;
  slices(n, &grid, &chunk);
  hipLaunchKernelGGL(padded_init_kernel, dim3(ctr_stream_grid(slots, kBlock)), dim3(kBlock), 0, st, send, inv, slots, n);
  CTR_REQUIRE(n == 0 || (ids && perm), CTR_EINVAL);
  hipLaunchKernelGGL(owner_count_kernel, dim3(grid), dim3(kBlock), 0, st, ids, n, chunk, world, vocab, cursor);
  hipLaunchKernelGGL(padded_place_kernel, dim3(grid), dim3(kBlock), 0, st, ids, n, chunk, world, vocab, cap, cursor, send,
                     perm, inv, state);
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
  CTR_REQUIRE(n == 0 || (ids && send && perm && inv), CTR_EINVAL);
  hipStream_t st = (hipStream_t)stream;
  int grid;
  int64_t chunk;
  slices(n, &grid, &chunk);
  hipLaunchKernelGGL(owner_count_kernel, dim3(grid), dim3(kBlock), 0, st, ids, n, chunk, world, vocab, cursor);
  hipLaunchKernelGGL(owner_place_kernel, dim3(grid), dim3(kBlock), 0, st, ids, n, chunk, world, vocab, cursor, counts, send,
                     perm, inv);
  return ctr_launch_status();
}
