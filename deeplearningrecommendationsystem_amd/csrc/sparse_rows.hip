// Opt-in sparse mode of the embedding backward / optimizer.  The reference's
// optimizer line (scripts/din.py:87: optim.Adam(model.parameters(), lr, weight_decay=1e-5))
// makes every step sweep whole tables: a dense (V,E) gradient is zero-filled, and Adam with L2
// reads and writes 7 x the table bytes (11.6 GB per step at 26 x 1e6 x 16) although a batch
// touches <= 65536 rows per table.  Here the dense gradient buffer becomes PERSISTENT
// accumulation scratch that is clean outside the rows a batch touched:
//   backward:   the unchanged scatter kernels add into it (no zero-fill), then ctr_rows_mark
//               appends every row touched for the first time to the table's unique-row list
//               (one flag word per row, one returning atomic per wave on the list length);
//   optimizer:  ctr_adam_rows updates p / exp_avg / exp_avg_sq of the listed rows only (L2 decay
//               applied to those rows: "lazy" decay), zeroes their gradient rows and flags again and
//               leaves the list empty -- every buffer is back to its clean state.
// No sort, no host sync, fixed launch geometry (hipGraph-capturable).  Semantics differ from the
// reference's dense Adam for untouched rows (no decay, no moment update there), hence opt-in.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;

struct MarkJob {
  const void* ids;      // int64 or float32 ids
  int64_t stride;       // elements between consecutive ids
  int64_t n;            // ids in this job
  int64_t vocab;
  int32_t* flags;       // [vocab], 0 = clean
  int32_t* rows;        // [cap] unique-row list
  int32_t* count;       // [1] list length
  int32_t is_float;
  int32_t cap;
};
struct MarkJobs {
  int n;
  MarkJob j[CTR_MAX_FIELDS];
};

__global__ void __launch_bounds__(kBlock) rows_mark_kernel(const MarkJobs J) {
  const MarkJob j = J.j[blockIdx.y];
  const int lane = threadIdx.x & 63;
  for (int64_t base = (int64_t)blockIdx.x * kBlock; base < j.n; base += (int64_t)gridDim.x * kBlock) {
    const int64_t i = base + threadIdx.x;
    int64_t r = -1;
    if (i < j.n)
      r = j.is_float ? (int64_t)ctr_ldg((const float*)j.ids + i * j.stride) : ctr_ldg((const int64_t*)j.ids + i * j.stride);
    bool first = false;
    if (r >= 0 && r < j.vocab) first = atomicExch(j.flags + r, 1) == 0;
    // one returning atomic per wave: the wave's first-touchers take consecutive list slots
    const unsigned long long mask = __ballot(first);
    if (mask != 0ull) {
      int slot0 = 0;
      if (lane == __ffsll((long long)mask) - 1) slot0 = atomicAdd(j.count, __popcll(mask));
      slot0 = __shfl(slot0, __ffsll((long long)mask) - 1, 64);
      if (first) {
        const int slot = slot0 + __popcll(mask & ((1ull << lane) - 1ull));
        if (slot < j.cap) j.rows[slot] = (int32_t)r;  // cap == vocab: cannot overflow (a row is listed once)
      }
    }
  }
}

struct RowsJob {
  float* param;
  float* grad;          // persistent (vocab, dim) accumulation buffer
  float* exp_avg;
  float* exp_avg_sq;
  int32_t* flags;
  int32_t* rows;
  int32_t* count;
  unsigned int* done;   // [1] ticket: the last workgroup of the job empties the list
  int32_t dim;
};
struct RowsJobs {
  int n;
  RowsJob j[CTR_MAX_FIELDS];
};

// MODE 0: Adam update of the listed rows; MODE 1: discard (zero_grad) -- both leave G / flags / list clean
template <int MODE>
__global__ void __launch_bounds__(kBlock)
rows_apply_kernel(const RowsJobs J, float lr, float beta2, float omb1, float omb2, float eps, float weight_decay,
                  float bc1, float bc2_sqrt) {
  const RowsJob j = J.j[blockIdx.y];
  const int n = *(volatile int32_t*)j.count;
  const int dim = j.dim;
  const float step_size = lr / bc1;
  // one lane per element, consecutive lanes on consecutive elements of a row (whole 4*dim-byte segments)
  const int64_t total = (int64_t)n * dim;
  for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < total; g += (int64_t)gridDim.x * kBlock) {
    const int64_t s = g / dim;
    const int e = (int)(g - s * dim);
    const int64_t r = j.rows[s];
    const int64_t off = r * dim + e;
    if (MODE == 0) {
      const float p = j.param[off];
      const float gr = fmaf(weight_decay, p, j.grad[off]);   // grad.add(param, alpha=wd), on touched rows only
      const float m = fmaf(omb1, gr - j.exp_avg[off], j.exp_avg[off]);
      const float v = j.exp_avg_sq[off] * beta2 + (omb2 * gr) * gr;
      j.exp_avg[off] = m;
      j.exp_avg_sq[off] = v;
      j.param[off] = p - step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
    }
    j.grad[off] = 0.0f;
    if (e == 0) j.flags[r] = 0;
  }
  // every workgroup has read `n` by now; the last one to finish empties the list
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(j.done, 1u) == gridDim.x - 1) {
      *j.count = 0;
      *j.done = 0u;
    }
  }
}

}  // namespace

extern "C" int ctr_rows_mark(const ctr_rows_mark_t* jobs, int njobs, void* stream) {
  CTR_REQUIRE(jobs && njobs >= 0 && njobs <= CTR_MAX_FIELDS, CTR_EINVAL);
  if (njobs == 0) return CTR_OK;
  MarkJobs J;
  J.n = njobs;
  int64_t longest = 0;
  for (int i = 0; i < njobs; ++i) {
    const ctr_rows_mark_t& a = jobs[i];
    CTR_REQUIRE(a.n >= 0 && a.vocab > 0 && a.vocab < (1ll << 31) && a.flags && a.rows && a.count, CTR_EINVAL);
    CTR_REQUIRE(a.n == 0 || (a.ids && a.stride >= 1), CTR_EINVAL);
    J.j[i] = MarkJob{a.ids, a.stride, a.n, a.vocab, a.flags, a.rows, a.count, a.ids_are_float, (int32_t)a.vocab};
    longest = a.n > longest ? a.n : longest;
  }
  if (longest == 0) return CTR_OK;
  int64_t gx = ctr_ceil_div(longest, kBlock);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(rows_mark_kernel, dim3((unsigned)gx, (unsigned)njobs), dim3(kBlock), 0, (hipStream_t)stream, J);
  return ctr_launch_status();
}

static int pack_rows(const ctr_rows_table_t* tables, int ntables, RowsJobs* J) {
  CTR_REQUIRE(tables && ntables >= 0 && ntables <= CTR_MAX_FIELDS, CTR_EINVAL);
  J->n = ntables;
  for (int i = 0; i < ntables; ++i) {
    const ctr_rows_table_t& t = tables[i];
    CTR_REQUIRE(t.grad && t.flags && t.rows && t.count && t.done && t.dim > 0, CTR_EINVAL);
    J->j[i] = RowsJob{t.param, t.grad, t.exp_avg, t.exp_avg_sq, t.flags, t.rows, t.count, t.done, t.dim};
  }
  return CTR_OK;
}

extern "C" int ctr_adam_rows(const ctr_rows_table_t* tables, int ntables, double lr, double beta1, double beta2,
                             double eps, double weight_decay, int64_t step, void* stream) {
  CTR_REQUIRE(step >= 1, CTR_EINVAL);
  RowsJobs J;
  int rc = pack_rows(tables, ntables, &J);
  if (rc != CTR_OK || ntables == 0) return rc;
  for (int i = 0; i < ntables; ++i) CTR_REQUIRE(tables[i].param && tables[i].exp_avg && tables[i].exp_avg_sq, CTR_EINVAL);
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  // fixed geometry (the list length lives on the device): enough workgroups for 65536 rows x 16 per table
  hipLaunchKernelGGL(rows_apply_kernel<0>, dim3(512u, (unsigned)ntables), dim3(kBlock), 0, (hipStream_t)stream, J,
                     (float)lr, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
                     (float)weight_decay, (float)bc1, (float)sqrt(bc2));
  return ctr_launch_status();
}

extern "C" int ctr_rows_discard(const ctr_rows_table_t* tables, int ntables, void* stream) {
  RowsJobs J;
  int rc = pack_rows(tables, ntables, &J);
  if (rc != CTR_OK || ntables == 0) return rc;
  hipLaunchKernelGGL(rows_apply_kernel<1>, dim3(512u, (unsigned)ntables), dim3(kBlock), 0, (hipStream_t)stream, J, 0.f,
                     0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f);
  return ctr_launch_status();
}
