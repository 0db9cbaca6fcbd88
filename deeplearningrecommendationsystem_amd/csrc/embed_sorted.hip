// Embedding backward for SMALL tables hit by a LARGE batch (ml-100k: 943 users / 1682
// items, batch 65536 -> every gradient row is the sum of ~40-70 samples).  Scattering
// sample by sample costs one fp32 atomic per element and the memory-side atomic units,
// not HBM, set the pace (~1.3 TB/s of added bytes, measured).  Here the samples are first
// bucketed by row (counting sort on the ids: histogram -> scan -> scatter), then lane
// groups walk the sorted list, sum each run of equal rows in registers and issue ONE
// atomic per run and chunk: ~13x fewer atomics, and the gout rows are read exactly once
// at the HBM rate.  This is the "segmented-reduce" half of the sparse backward; the
// atomic kernel in embed.hip stays the path for large tables, where runs have length 1.
//
// A sort is shared by every table addressed through the same id column (NeuralCF: the
// MLP and GMF user tables use one sort of the user ids).
#include "ctr_common.h"
#include <stdlib.h>

namespace {

constexpr int kBlock = 256;
constexpr int kMaxJobs = 8;
constexpr int kMaxStreams = 16;
constexpr int kMaxVocab = 8192;  // LDS histogram / one-workgroup scan
constexpr int kRunDefault = 8;   // sorted samples per lane group

struct SortJob {
  const int64_t* idx;   // ID_I64 / PROD: ids, else NULL
  int64_t idx_stride;
  const float* xcol;    // ID_F32: x + src_col
  int64_t ldx;
  int32_t vocab;
  int32_t* hist;        // [nblk][vocab] per-workgroup histograms
  int32_t* base;        // [nblk][vocab] samples of the row in earlier workgroups
  int32_t* total;       // [vocab] samples of the row
  int32_t* order;       // [batch] sample of every sorted position
  int32_t* keys;        // [batch] row of every sorted position
  // companion column (a product field's other factor): its row per sorted position, written by the scatter pass so
  // that the reduce pass needs no dependent id load per sample
  const int64_t* cidx;
  int64_t cidx_stride;
  int64_t cvocab;
  int32_t* ckeys;       // [batch] or NULL
};
struct SortJobs {
  int n;
  SortJob j[kMaxJobs];
};
struct Stream {
  int job;
  int width, out_col, lpr;
  float* grad;
  const float* ptable;  // PROD: the other factor's table, else NULL
  const int64_t* pidx;
  int64_t pidx_stride;
  int64_t pvocab;
  int use_ckeys;        // the partner rows are the job's companion column
};
struct Streams {
  int n;
  Stream s[kMaxStreams];
};

__device__ __forceinline__ int load_row(const SortJob& j, uint32_t b) {
  int64_t r = j.idx ? ctr_ldg(j.idx + (int64_t)b * j.idx_stride) : (int64_t)ctr_ldg(j.xcol + (int64_t)b * j.ldx);
  if (r < 0 || r >= j.vocab) r = 0;  // same clamp as the forward (the flag was raised there)
  return (int)r;
}

// Counting sort without a single global atomic (65536 returning atomics on 943 addresses
// serialise into ~70-long chains: 17 us measured).  Workgroup w owns the contiguous slice
// [w*chunk, (w+1)*chunk) of the batch in both passes:
//   count:   LDS histogram of the slice -> hist[w][:]            (plain stores, no memset)
//   colscan: base[w][v] = sum_{w'<w} hist[w'][v],  total[v] = sum_w hist[w][v]
//            (folding this pass into the scatter workgroups was tried: each then sums nblk histograms itself, a
//            chain of L2 round trips that took 34 us against 6 + 7 us for the two launches)
//   scatter: LDS cursors start at (exclusive scan of total)[:] + base[w][:], ds_add_rtn hands
//            out the positions
__global__ void __launch_bounds__(kBlock) sort_count_kernel(const SortJobs J, uint32_t batch, uint32_t chunk) {
  extern __shared__ int s_hist[];
  const SortJob j = J.j[blockIdx.y];
  for (int v = threadIdx.x; v < j.vocab; v += kBlock) s_hist[v] = 0;
  __syncthreads();
  const uint32_t lo = blockIdx.x * chunk, hi = lo + chunk < batch ? lo + chunk : batch;
  for (uint32_t b = lo + threadIdx.x; b < hi; b += kBlock) atomicAdd(&s_hist[load_row(j, b)], 1);
  __syncthreads();
  int32_t* out = j.hist + (int64_t)blockIdx.x * j.vocab;
  for (int v = threadIdx.x; v < j.vocab; v += kBlock) out[v] = s_hist[v];
}

// column pass: thread = row v (coalesced over v), exclusive prefix over the workgroups w
__global__ void __launch_bounds__(kBlock) sort_colscan_kernel(const SortJobs J, int nblk) {
  const SortJob j = J.j[blockIdx.y];
  const int v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= j.vocab) return;
  const int32_t* __restrict__ hist = j.hist;
  int32_t* __restrict__ base = j.base;
  int run = 0;
  for (int w0 = 0; w0 < nblk; w0 += 8) {
    int t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = w0 + u < nblk ? hist[(int64_t)(w0 + u) * j.vocab + v] : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (w0 + u < nblk) base[(int64_t)(w0 + u) * j.vocab + v] = run;
      run += t[u];
    }
  }
  j.total[v] = run;
}

// every workgroup scans the row totals itself (<= 32 per thread + one 256-wide block scan:
// cheaper than a separate single-workgroup launch), then hands out positions from LDS cursors
__global__ void __launch_bounds__(kBlock) sort_scatter_kernel(const SortJobs J, uint32_t batch, uint32_t chunk) {
  extern __shared__ int s_cur[];
  __shared__ int s_part[kBlock];
  const SortJob j = J.j[blockIdx.y];
  const int per = (j.vocab + kBlock - 1) / kBlock;  // consecutive rows owned by a thread
  const int v0 = threadIdx.x * per;
  int sum = 0;
  for (int i = 0; i < per; ++i) {
    const int v = v0 + i;
    const int t = v < j.vocab ? j.total[v] : 0;
    if (v < j.vocab) s_cur[v] = sum;  // exclusive prefix inside the thread's run
    sum += t;
  }
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < kBlock; d <<= 1) {
    const int t = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
    __syncthreads();
    s_part[threadIdx.x] += t;
    __syncthreads();
  }
  const int before = s_part[threadIdx.x] - sum;
  const int32_t* mine = j.base + (int64_t)blockIdx.x * j.vocab;
  for (int i = 0; i < per; ++i) {
    const int v = v0 + i;
    if (v < j.vocab) s_cur[v] += before + mine[v];
  }
  __syncthreads();
  const uint32_t lo = blockIdx.x * chunk, hi = lo + chunk < batch ? lo + chunk : batch;
  for (uint32_t b = lo + threadIdx.x; b < hi; b += kBlock) {
    const int r = load_row(j, b);
    int64_t c = j.cidx ? ctr_ldg(j.cidx + (int64_t)b * j.cidx_stride) : 0;
    if (c < 0 || c >= j.cvocab) c = 0;
    const int pos = atomicAdd(&s_cur[r], 1);
    j.order[pos] = (int)b;
    j.keys[pos] = r;
    if (j.cidx) j.ckeys[pos] = (int)c;
  }
}

__device__ __forceinline__ void flush_run(float* grad, int row, int width, int c, const float4& acc) {
#ifdef CTR_SEG_NOATOMIC
  if (acc.x == 1234.5f) grad[0] = acc.y;
  return;
#endif
  float* p = grad + (int64_t)row * width + c;
  ctr_atomic_add_global(p + 0, acc.x);
  ctr_atomic_add_global(p + 1, acc.y);
  ctr_atomic_add_global(p + 2, acc.z);
  ctr_atomic_add_global(p + 3, acc.w);
}

// blockIdx.y = stream.  `lpr` lanes (width / 4 rounded up to a power of two) share a row, each
// holding one dwordx4; a group owns kRun consecutive sorted positions, all rows in flight.
// Runs that lie strictly inside a group's chunk are complete rows: one uncontended atomic.
// The first and the last run of a chunk may continue in the neighbouring groups; those
// partials go through LDS and the workgroup (256 * kRun / lpr ... consecutive positions)
// merges neighbours with equal rows before it touches memory, so a hot row with thousands
// of samples costs one atomic per workgroup, not one per group (same-address fp32 atomics
// serialise: 190 of them on the hottest ml-100k row were 20 us of this kernel's 35).
template <int kRun>
__global__ void __launch_bounds__(kBlock)
seg_reduce_kernel(const SortJobs J, const Streams T, uint32_t batch, const float* __restrict__ gout, int64_t ldo) {
  __shared__ float s_sum[2 * kBlock * 4];  // [head|tail][group][lpr * 4]
  __shared__ int s_key[2 * kBlock];        // [head|tail][group], -1 = none
  const Stream st = T.s[blockIdx.y];
  const SortJob j = J.j[st.job];
  const int sub = threadIdx.x % st.lpr, grp = threadIdx.x / st.lpr, ngroups = kBlock / st.lpr;
  const uint32_t gid = blockIdx.x * ngroups + grp;
  const bool active = (uint64_t)gid * kRun < batch;
  const uint32_t p0 = active ? gid * kRun : 0;
  const int n = !active ? 0 : (batch - p0 < (uint32_t)kRun ? (int)(batch - p0) : kRun);
  const int c = sub * 4;
  const bool live = c < st.width;
  int head_key = -1, tail_key = -1;
  float4 head = make_float4(0.f, 0.f, 0.f, 0.f), acc = head;
  if (active) {
    // every load of the chunk is issued before the first use: keys/order, then the kRun
    // gradient rows (and the partner rows), then one serial pass over registers
    int kk[kRun], bb[kRun], cc[kRun];
    float4 g[kRun];
    const bool use_c = st.ptable && st.use_ckeys;
#pragma unroll
    for (int u = 0; u < kRun; ++u) {
      const uint32_t p = u < n ? p0 + u : p0;
      kk[u] = j.keys[p];
      bb[u] = j.order[p];
      cc[u] = use_c ? j.ckeys[p] : 0;
    }
#pragma unroll
    for (int u = 0; u < kRun; ++u) {
      g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) g[u] = ctr_ldg(reinterpret_cast<const float4*>(gout + (int64_t)bb[u] * ldo + st.out_col + c));
    }
    if (st.ptable) {
      // d(t1 * t2) = g * t2: the partner row of the same sample (tables this small sit in L2)
      int64_t pr[kRun];
#pragma unroll
      for (int u = 0; u < kRun; ++u) {
        pr[u] = cc[u];
        if (!use_c) {
          pr[u] = ctr_ldg(st.pidx + (int64_t)bb[u] * st.pidx_stride);
          if (pr[u] < 0 || pr[u] >= st.pvocab) pr[u] = 0;
        }
      }
      if (live) {
#pragma unroll
        for (int u = 0; u < kRun; ++u) {
          const float4 t = ctr_ldg(reinterpret_cast<const float4*>(st.ptable + pr[u] * st.width + c));
          g[u].x *= t.x; g[u].y *= t.y; g[u].z *= t.z; g[u].w *= t.w;
        }
      }
    }
    int cur = kk[0];
    bool first = true;
#pragma unroll
    for (int u = 0; u < kRun; ++u) {
      if (u < n) {
        if (kk[u] != cur) {
          if (first) {
            head_key = cur;
            head = acc;
            first = false;
          } else if (live) {
            flush_run(st.grad, cur, st.width, c, acc);
          }
          acc = make_float4(0.f, 0.f, 0.f, 0.f);
          cur = kk[u];
        }
        acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w;
      }
    }
    if (first) {
      head_key = cur;
      head = acc;
    } else {
      tail_key = cur;  // acc holds the last run
    }
  }
  const int lw = st.lpr * 4;
  *reinterpret_cast<float4*>(s_sum + (0 * ngroups + grp) * lw + c) = head;
  *reinterpret_cast<float4*>(s_sum + (1 * ngroups + grp) * lw + c) = acc;
  if (sub == 0) {
    s_key[grp] = head_key;
    s_key[ngroups + grp] = tail_key;
  }
  __syncthreads();
  if (threadIdx.x < st.lpr && live) {
    int cur = -1;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < 2 * ngroups; ++i) {
      const int slot = i & 1, gi = i >> 1;
      const int key = s_key[slot * ngroups + gi];
      if (key < 0) continue;
      if (key != cur) {
        if (cur >= 0) flush_run(st.grad, cur, st.width, c, sum);
        sum = make_float4(0.f, 0.f, 0.f, 0.f);
        cur = key;
      }
      const float4 t = *reinterpret_cast<const float4*>(s_sum + (slot * ngroups + gi) * lw + c);
      sum.x += t.x; sum.y += t.y; sum.z += t.z; sum.w += t.w;
    }
    if (cur >= 0) flush_run(st.grad, cur, st.width, c, sum);
  }
}

inline int pow2_ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

struct Key {
  const void* base;
  int64_t stride, vocab;
};

}  // namespace

// Takes the id / product fields whose tables are small relative to the batch.  On return
// handled[i] is 1 for a field taken completely, 2 / 3 for a PROD field whose first / second
// factor alone was taken (never happens today: both or none), 0 otherwise.  The sort buffers
// are carved from the END of the workspace (embed.hip's bag partials use its start).
int ctr_embed_bwd_sorted(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx, int64_t batch,
                         const float* gout, int64_t ldo, float* workspace, int64_t workspace_floats,
                         int64_t* used_floats, unsigned char* handled, hipStream_t st) {
  *used_floats = 0;
  for (int i = 0; i < nfields; ++i) handled[i] = 0;
  if (!workspace || batch >= (1ll << 31) || ldo % 4 != 0) return CTR_OK;
  if (!ctr_aligned16(gout)) return CTR_OK;
  SortJobs J;
  Streams T;
  J.n = 0;
  T.n = 0;
  Key keys[kMaxJobs];
  int vstreams = 0;  // reduce streams the fields would get with every gradient present (limits the job set)
  auto job_of = [&](const void* base, int64_t stride, int64_t vocab) -> int {
    for (int k = 0; k < J.n; ++k)
      if (keys[k].base == base && keys[k].stride == stride && keys[k].vocab == vocab) return k;
    if (J.n == kMaxJobs) return -1;
    keys[J.n] = Key{base, stride, vocab};
    return J.n++;
  };
  auto small = [&](int64_t vocab) { return vocab <= kMaxVocab && batch >= 8 * vocab; };
  for (int i = 0; i < nfields; ++i) {
    const ctr_field_t& f = fields[i];
    const bool shape_ok = f.width % 4 == 0 && f.out_col % 4 == 0 && f.width <= 256;
    if (!shape_ok) continue;
    const bool g1 = f.grad && ctr_aligned16(f.grad), g2 = f.grad2 && ctr_aligned16(f.grad2);
    if ((f.kind == CTR_FIELD_ID_I64 || f.kind == CTR_FIELD_ID_F32) && g1 && small(f.vocab)) {
      if (vstreams + 1 > kMaxStreams) continue;
      const int before = J.n;
      const int jb = f.kind == CTR_FIELD_ID_I64 ? job_of(f.idx, f.idx_stride, f.vocab)
                                                : job_of(x + f.src_col, ldx, f.vocab);
      if (jb < 0) continue;
      if (jb == before) {
        SortJob& sj = J.j[jb];
        sj = SortJob{};
        sj.idx = f.kind == CTR_FIELD_ID_I64 ? f.idx : nullptr;
        sj.idx_stride = f.idx_stride;
        sj.xcol = f.kind == CTR_FIELD_ID_F32 ? x + f.src_col : nullptr;
        sj.ldx = ldx;
        sj.vocab = (int32_t)f.vocab;
      }
      vstreams += 1;
      if (g1) {
        T.s[T.n++] = Stream{jb, f.width, f.out_col, pow2_ceil(f.width / 4), f.grad, nullptr, nullptr, 0, 0, 0};
        handled[i] = 1;
      }
    } else if (f.kind == CTR_FIELD_PROD_I64 && g1 && g2 && small(f.vocab) && small(f.vocab2) &&
               ctr_aligned16(f.table) && ctr_aligned16(f.table2)) {
      if (vstreams + 2 > kMaxStreams) continue;
      const int n0 = J.n;
      const int j1 = job_of(f.idx, f.idx_stride, f.vocab);
      const int n1 = J.n;
      const int j2 = j1 < 0 ? -1 : job_of(f.idx2, f.idx_stride, f.vocab2);
      if (j1 < 0 || j2 < 0) {
        J.n = n0;  // do not keep a job nobody uses
        continue;
      }
      if (j1 == n0) {
        J.j[j1] = SortJob{};
        J.j[j1].idx = f.idx;
        J.j[j1].idx_stride = f.idx_stride;
        J.j[j1].vocab = (int32_t)f.vocab;
      }
      if (j2 == n1 && j2 != j1) {
        J.j[j2] = SortJob{};
        J.j[j2].idx = f.idx2;
        J.j[j2].idx_stride = f.idx_stride;
        J.j[j2].vocab = (int32_t)f.vocab2;
      }
      // the other factor's id column rides along as the job's companion (the first product field of a job wins)
      auto companion = [&](int jb, const int64_t* cidx, int64_t cvocab) -> int {
        SortJob& sj = J.j[jb];
        if (!sj.cidx && cvocab < (1ll << 31)) {
          sj.cidx = cidx;
          sj.cidx_stride = f.idx_stride;
          sj.cvocab = cvocab;
        }
        return sj.cidx == cidx && sj.cidx_stride == f.idx_stride && sj.cvocab == cvocab;
      };
      const int lpr = pow2_ceil(f.width / 4);
      const int c1 = companion(j1, f.idx2, f.vocab2), c2 = companion(j2, f.idx, f.vocab);
      vstreams += 2;
      if (g1 && g2) {
        T.s[T.n++] = Stream{j1, f.width, f.out_col, lpr, f.grad, f.table2, f.idx2, f.idx_stride, f.vocab2, c1};
        T.s[T.n++] = Stream{j2, f.width, f.out_col, lpr, f.grad2, f.table, f.idx, f.idx_stride, f.vocab, c2};
        handled[i] = 1;
      }
    }
  }
  if (J.n == 0 || T.n == 0) {
    for (int i = 0; i < nfields; ++i) handled[i] = 0;
    return CTR_OK;
  }
  // workgroups of the two sort passes: contiguous slices of >= 1024 samples
  int nblk = (int)ctr_ceil_div(batch, 1024);
  if (nblk > 128) nblk = 128;
  const int64_t chunk = ctr_ceil_div(batch, nblk);
  nblk = (int)ctr_ceil_div(batch, chunk);
  // int32 buffers at the end of the workspace: per job (2 * nblk + 1) * vocab + 2 (3 with a companion column) * batch
  const int64_t b4 = (batch + 3) / 4 * 4;
  int64_t need = 0;
  for (int k = 0; k < J.n; ++k)
    need += 2 * (((int64_t)nblk * J.j[k].vocab + 3) / 4 * 4) + (J.j[k].vocab + 3) / 4 * 4 + (J.j[k].cidx ? 3 : 2) * b4;
  if (need > workspace_floats) {
    for (int i = 0; i < nfields; ++i) handled[i] = 0;
    return CTR_OK;
  }
  int32_t* at = reinterpret_cast<int32_t*>(workspace + (workspace_floats - need));
  int maxv = 0;
  for (int k = 0; k < J.n; ++k) {
    const int64_t hv = ((int64_t)nblk * J.j[k].vocab + 3) / 4 * 4;
    J.j[k].hist = at;
    at += hv;
    J.j[k].base = at;
    at += hv;
    J.j[k].total = at;
    at += (J.j[k].vocab + 3) / 4 * 4;
    J.j[k].order = at;
    at += b4;
    J.j[k].keys = at;
    at += b4;
    J.j[k].ckeys = nullptr;
    if (J.j[k].cidx) {
      J.j[k].ckeys = at;
      at += b4;
    }
    maxv = J.j[k].vocab > maxv ? J.j[k].vocab : maxv;
  }
  *used_floats = need;
  hipLaunchKernelGGL(sort_count_kernel, dim3(nblk, J.n), dim3(kBlock), sizeof(int) * maxv, st, J, (uint32_t)batch,
                     (uint32_t)chunk);
  hipLaunchKernelGGL(sort_colscan_kernel, dim3((unsigned)ctr_ceil_div(maxv, kBlock), J.n), dim3(kBlock), 0, st, J, nblk);
  hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblk, J.n), dim3(kBlock), sizeof(int) * maxv, st, J, (uint32_t)batch,
                     (uint32_t)chunk);
  // grid.x sized for the narrowest stream's groups-per-workgroup; wider streams exit early
  int maxlpr = 1;
  for (int k = 0; k < T.n; ++k) maxlpr = T.s[k].lpr > maxlpr ? T.s[k].lpr : maxlpr;
  // sorted samples per lane group: 8 (A/B on MI355X, NeuralCF step: 16 -> 181.3 us, 8 -> 177.2 us, 4 -> 177.6 us)
  constexpr int run = kRunDefault;
  const int64_t groups = ctr_ceil_div(batch, run);
  const int gx = (int)ctr_ceil_div(groups, kBlock / maxlpr);
  const dim3 grid(gx, T.n);
  hipLaunchKernelGGL(seg_reduce_kernel<run>, grid, dim3(kBlock), 0, st, J, T, (uint32_t)batch, gout, ldo);
  return ctr_launch_status();
}
