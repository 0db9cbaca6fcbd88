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
    // plain look first: a hot row (the sequences' padding id is a quarter of all DIN history positions) would
    // otherwise take one same-address atomic per occurrence (~60 ns each, serialised: +5 ms on DIN cfg5).
    // A stale 0 only costs the atomic it was meant to save; a 1 can only have been written in this epoch.
    if (r >= 0 && r < j.vocab && *(volatile const int32_t*)(j.flags + r) == 0) first = atomicExch(j.flags + r, 1) == 0;
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

struct AdamConsts {
  float step_size, beta2, omb1, omb2, eps, weight_decay, bc2_sqrt;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamConsts& c) {
  const float gr = fmaf(c.weight_decay, p, g);   // grad.add(param, alpha=wd), on touched rows only
  m = fmaf(c.omb1, gr - m, m);                   // exp_avg.lerp_(grad, 1 - beta1)
  v = v * c.beta2 + (c.omb2 * gr) * gr;          // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  p -= c.step_size * (m / (sqrtf(v) / c.bc2_sqrt + c.eps));
}

// MODE 0: Adam update of the listed rows; MODE 1: discard (zero_grad) -- both leave G / flags / list clean.
// VEC 4: dim % 4 == 0 and 16-byte aligned rows, dim/4 lanes per row (one dwordx4 of p, g, m, v each);
// VEC 1: any dim, one lane per element.
template <int MODE, int VEC>
__global__ void __launch_bounds__(kBlock)
rows_apply_kernel(const RowsJobs J, const AdamConsts c) {
  const RowsJob j = J.j[blockIdx.y];
  const int n = *(volatile int32_t*)j.count;
  const uint32_t upr = (uint32_t)(j.dim / VEC);          // lanes per row
  const uint64_t total = (uint64_t)n * upr;
  const bool pow2 = (upr & (upr - 1)) == 0;
  const int shift = 31 - __clz((int)upr);
  for (uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x; g < total; g += (uint64_t)gridDim.x * kBlock) {
    const uint64_t s = pow2 ? g >> shift : g / upr;
    const uint32_t u = (uint32_t)(g - s * upr);
    const int64_t r = j.rows[s];
    const int64_t off = r * j.dim + (int64_t)u * VEC;
    if (VEC == 4) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      if (MODE == 0) {
        f4 p = *(CTR_GLOBAL f4*)(j.param + off), gg = *(const CTR_GLOBAL f4*)(j.grad + off);
        f4 m = *(CTR_GLOBAL f4*)(j.exp_avg + off), v = *(CTR_GLOBAL f4*)(j.exp_avg_sq + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pe = p[e], me = m[e], ve = v[e];
          adam_elem(pe, gg[e], me, ve, c);
          p[e] = pe; m[e] = me; v[e] = ve;
        }
        *(CTR_GLOBAL f4*)(j.exp_avg + off) = m;
        *(CTR_GLOBAL f4*)(j.exp_avg_sq + off) = v;
        *(CTR_GLOBAL f4*)(j.param + off) = p;
      }
      const f4 z = {0.f, 0.f, 0.f, 0.f};
      *(CTR_GLOBAL f4*)(j.grad + off) = z;
    } else {
      if (MODE == 0) {
        float p = j.param[off], m = j.exp_avg[off], v = j.exp_avg_sq[off];
        adam_elem(p, j.grad[off], m, v, c);
        j.exp_avg[off] = m;
        j.exp_avg_sq[off] = v;
        j.param[off] = p;
      }
      j.grad[off] = 0.0f;
    }
    if (u == 0) j.flags[r] = 0;
  }
  // The last workgroup to finish empties the list.  A workgroup takes its ticket after its loop, whose bounds
  // came from `n`: every workgroup has read the length before the reset can happen.  No fence: nothing this
  // kernel wrote is read by another workgroup (a __threadfence() here wrote the L2's dirty lines back once per
  // workgroup -- 13312 times over 1.3 GB of row updates: 1.10 ms instead of ~0.25).
  __syncthreads();
  if (threadIdx.x == 0) {
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

// tables are launched in two groups: those whose rows can be handled as dwordx4 units, and the rest
static int pack_rows(const ctr_rows_table_t* tables, int ntables, RowsJobs* J4, RowsJobs* J1) {
  CTR_REQUIRE(tables && ntables >= 0 && ntables <= CTR_MAX_FIELDS, CTR_EINVAL);
  J4->n = J1->n = 0;
  for (int i = 0; i < ntables; ++i) {
    const ctr_rows_table_t& t = tables[i];
    CTR_REQUIRE(t.grad && t.flags && t.rows && t.count && t.done && t.dim > 0, CTR_EINVAL);
    const bool v4 = t.dim % 4 == 0 && ctr_aligned16(t.grad) && (!t.param || ctr_aligned16(t.param)) &&
                    (!t.exp_avg || ctr_aligned16(t.exp_avg)) && (!t.exp_avg_sq || ctr_aligned16(t.exp_avg_sq));
    RowsJobs* J = v4 ? J4 : J1;
    J->j[J->n++] = RowsJob{t.param, t.grad, t.exp_avg, t.exp_avg_sq, t.flags, t.rows, t.count, t.done, t.dim};
  }
  return CTR_OK;
}

template <int MODE>
static int launch_rows(const RowsJobs& J4, const RowsJobs& J1, const AdamConsts& c, hipStream_t st) {
  // fixed geometry (the list length lives on the device)
  if (J4.n) hipLaunchKernelGGL((rows_apply_kernel<MODE, 4>), dim3(512u, (unsigned)J4.n), dim3(kBlock), 0, st, J4, c);
  if (J1.n) hipLaunchKernelGGL((rows_apply_kernel<MODE, 1>), dim3(512u, (unsigned)J1.n), dim3(kBlock), 0, st, J1, c);
  return ctr_launch_status();
}

extern "C" int ctr_adam_rows(const ctr_rows_table_t* tables, int ntables, double lr, double beta1, double beta2,
                             double eps, double weight_decay, int64_t step, void* stream) {
  CTR_REQUIRE(step >= 1, CTR_EINVAL);
  RowsJobs J4, J1;
  int rc = pack_rows(tables, ntables, &J4, &J1);
  if (rc != CTR_OK || ntables == 0) return rc;
  for (int i = 0; i < ntables; ++i) CTR_REQUIRE(tables[i].param && tables[i].exp_avg && tables[i].exp_avg_sq, CTR_EINVAL);
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const AdamConsts c{(float)(lr / bc1), (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
                     (float)weight_decay, (float)sqrt(bc2)};
  return launch_rows<0>(J4, J1, c, (hipStream_t)stream);
}

extern "C" int ctr_rows_discard(const ctr_rows_table_t* tables, int ntables, void* stream) {
  RowsJobs J4, J1;
  int rc = pack_rows(tables, ntables, &J4, &J1);
  if (rc != CTR_OK || ntables == 0) return rc;
  return launch_rows<1>(J4, J1, AdamConsts{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f}, (hipStream_t)stream);
}
