// Backward of the weighted bag poolings ("multi-hot matmul embeddings", K2): the gradient
// of a K-row bag table is  dT[j, :] = sum_b x[b, a+j] * g[b, cols]  -- an (K x B)(B x E)
// product whose K*E outputs are shared by the whole batch.  Adding sample by sample needs
// atomics on <= 21 rows (global: serialised; LDS: ds_add_f32 runs at ~200 cycles per
// wave-instruction, measured -- FFM's 8 bag tables cost 250 us that way).  Here nothing is
// added atomically: a lane owns ONE output column and keeps all K partial sums of it in
// registers; a wave covers 64/E samples x E columns per step, so a step is one coalesced
// load of g, K broadcast loads of x and K FMAs.  Lanes of equal column are combined by
// shuffles, the waves of a workgroup through LDS, the workgroups through the workspace
// (reduce.hip, fixed order -> reproducible).
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBags = 16;

struct Bag {
  const float* xcol;  // x + src_col
  int out_col;        // first column of g
  int width;          // E (power of two, <= 64)
  int rows;           // K
  int slab_off;       // first float of this bag inside a workgroup's partial
};
struct Bags {
  int n;
  Bag b[kMaxBags];
};

// blockIdx.y = bag, blockIdx.x = slice of the batch.  KMAX >= rows.
template <int KMAX>
__global__ void __launch_bounds__(kBlock)
bag_bwd_kernel(const Bags B, int first, int64_t ldx, uint32_t batch, const float* __restrict__ gout, int64_t ldo,
               float* __restrict__ ws, int64_t slab) {
  __shared__ float s_part[(kBlock / 64) * KMAX * 64];
  const Bag bag = B.b[first + blockIdx.y];
  if ((int)blockIdx.z * 64 >= bag.width && blockIdx.z > 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // rows wider than a wave are cut into 64-column chunks (blockIdx.z)
  const int e = bag.width < 64 ? bag.width : 64, spw = 64 / e;  // columns / samples per wave and step
  const int cbase = blockIdx.z * 64;
  const int col = lane & (e - 1), sub = lane / e;
  float acc[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) acc[j] = 0.0f;
  const uint32_t step = gridDim.x * (kBlock / 64) * spw;
  for (uint32_t b0 = (blockIdx.x * (kBlock / 64) + wave) * spw; b0 < batch; b0 += 2 * step) {
    // two samples per lane in flight
    const uint32_t s0 = b0 + sub, s1 = b0 + step + sub;
    const bool ok0 = s0 < batch, ok1 = s1 < batch;
    const float g0 = ok0 ? ctr_ldg(gout + (int64_t)s0 * ldo + bag.out_col + cbase + col) : 0.0f;
    const float g1 = ok1 ? ctr_ldg(gout + (int64_t)s1 * ldo + bag.out_col + cbase + col) : 0.0f;
    const float* x0 = bag.xcol + (int64_t)(ok0 ? s0 : 0) * ldx;
    const float* x1 = bag.xcol + (int64_t)(ok1 ? s1 : 0) * ldx;
    float w0[KMAX], w1[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      w0[j] = j < bag.rows ? ctr_ldg(x0 + j) : 0.0f;
      w1[j] = j < bag.rows ? ctr_ldg(x1 + j) : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < KMAX; ++j) acc[j] = fmaf(w1[j], g1, fmaf(w0[j], g0, acc[j]));
  }
  // lanes holding the same column (different samples of the step)
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    for (int o = e; o < 64; o <<= 1) acc[j] += __shfl_xor(acc[j], o, 64);
  if (sub == 0) {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) s_part[(wave * KMAX + j) * 64 + col] = acc[j];
  }
  __syncthreads();
  float* out = ws + (int64_t)blockIdx.x * slab + bag.slab_off;
  for (int i = threadIdx.x; i < bag.rows * e; i += kBlock) {
    const int j = i / e, c = i - j * e;
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) t += s_part[(w * KMAX + j) * 64 + c];
    out[j * bag.width + cbase + c] = t;
  }
}

// ---- forward: out[b, cols] = sum_j x[b, a+j] * T[j, :] (model "multi-hot matmul embeddings").  A kernel of
// its own: inside the gather kernel a load-use loop over the K rows made a 641-wide stage with two 18 / 21-row
// bags run at 13 % of the HBM rate, and unrolling it THERE took registers (= resident waves) from the id
// gathers.  A thread owns 4 output floats of one bag of one sample, requests four table rows at a time and
// keeps the FMAs in row order (a one-hot slice returns the selected row bit-exactly, as the reference's
// matmul does).
struct BagF {
  const float* table;
  const float* xcol;  // x + src_col
  int out_col, width, rows, unit0;
};
struct BagsF {
  int n, units;  // float4 units per sample over all bags
  BagF b[kMaxBags];
};

__global__ void __launch_bounds__(kBlock)
bag_fwd_kernel(const BagsF B, int64_t ldx, uint32_t batch, float* __restrict__ out, int64_t ldo, const CtrFastDiv div,
               int vec_store /* output rows 16-byte aligned; else four 4-byte stores (ldo = 641: odd) */) {
  const uint32_t total = batch * (uint32_t)B.units;
  for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < total; g += gridDim.x * kBlock) {
    const uint32_t b = ctr_div(g, div);
    const int u = (int)(g - b * (uint32_t)B.units);
    int k = 0;
    while (k + 1 < B.n && u >= B.b[k + 1].unit0) ++k;
    const BagF bag = B.b[k];
    const int off = (u - bag.unit0) * 4;
    const float* xr = bag.xcol + (int64_t)b * ldx;
    const float* tab = bag.table + off;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int j = 0;
    for (; j + 4 <= bag.rows; j += 4) {
      float w[4];
      float4 t[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        w[q] = ctr_ldg(xr + j + q);
        t[q] = ctr_ldg(reinterpret_cast<const float4*>(tab + (int64_t)(j + q) * bag.width));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        v = make_float4(fmaf(w[q], t[q].x, v.x), fmaf(w[q], t[q].y, v.y), fmaf(w[q], t[q].z, v.z), fmaf(w[q], t[q].w, v.w));
    }
    for (; j < bag.rows; ++j) {
      const float w = ctr_ldg(xr + j);
      const float4 t = ctr_ldg(reinterpret_cast<const float4*>(tab + (int64_t)j * bag.width));
      v = make_float4(fmaf(w, t.x, v.x), fmaf(w, t.y, v.y), fmaf(w, t.z, v.z), fmaf(w, t.w, v.w));
    }
    float* o = out + (int64_t)b * ldo + bag.out_col + off;
    if (vec_store) {
      ctr_stg(reinterpret_cast<float4*>(o), v);
    } else {
      ctr_stg(o, v.x); ctr_stg(o + 1, v.y); ctr_stg(o + 2, v.z); ctr_stg(o + 3, v.w);
    }
  }
}

}  // namespace

// Forward of the bag fields with 16-byte aligned table rows and >= 32 columns under a batch >= 4096 (handled[i] = 1
// for those); the descriptors were validated by the caller (make_plan).
int ctr_embed_fwd_bags(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx, int64_t batch, float* out,
                       int64_t ldo, unsigned char* handled, hipStream_t st) {
  for (int i = 0; i < nfields; ++i) handled[i] = 0;
  if (!x || batch < 4096) return CTR_OK;
  bool vec_store = ctr_aligned16(out) && ldo % 4 == 0;
  BagsF B;
  B.n = 0;
  B.units = 0;
  for (int i = 0; i < nfields && B.n < kMaxBags; ++i) {
    const ctr_field_t& f = fields[i];
    if (f.kind != CTR_FIELD_BAG || f.width < 32 || f.width % 4 || !ctr_aligned16(f.table)) continue;
    vec_store = vec_store && f.out_col % 4 == 0;
    B.b[B.n++] = BagF{f.table, x + f.src_col, f.out_col, f.width, f.bag_size, B.units};
    B.units += f.width / 4;
    handled[i] = 1;
  }
  if (B.n == 0) return CTR_OK;
  if (batch * B.units >= (1ll << 32)) {
    for (int i = 0; i < nfields; ++i) handled[i] = 0;
    return CTR_OK;
  }
  const CtrFastDiv div = ctr_fastdiv((uint32_t)B.units);
  hipLaunchKernelGGL(bag_fwd_kernel, dim3(ctr_stream_grid(batch * B.units, kBlock)), dim3(kBlock), 0, st, B, ldx,
                     (uint32_t)batch, out, ldo, div, vec_store ? 1 : 0);
  return ctr_launch_status();
}

// Takes every bag field with a power-of-two width <= 1024 and <= 32 rows; handled[i] = 1 for
// those.  Uses the first *used_floats of the workspace.
int ctr_embed_bwd_bags(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx, int64_t batch,
                       const float* gout, int64_t ldo, float* workspace, int64_t workspace_floats,
                       int64_t* used_floats, unsigned char* handled, hipStream_t st) {
  *used_floats = 0;
  if (!workspace || !x || batch >= (1ll << 31)) return CTR_OK;
  Bags B;
  B.n = 0;
  int idx_of[kMaxBags];
  int64_t slab = 0;
  for (int i = 0; i < nfields && B.n < kMaxBags; ++i) {
    const ctr_field_t& f = fields[i];
    if (f.kind != CTR_FIELD_BAG || !f.grad || handled[i]) continue;
    const bool pow2 = f.width > 0 && (f.width & (f.width - 1)) == 0;
    if (!pow2 || f.width > 1024 || f.bag_size > 32) continue;
    B.b[B.n] = Bag{x + f.src_col, f.out_col, f.width, f.bag_size, (int)slab};
    idx_of[B.n++] = i;
    slab += (int64_t)f.bag_size * f.width;
  }
  if (B.n == 0) return CTR_OK;
  // ~128 samples per workgroup: enough waves in flight to hide the load latency of the short
  // per-wave loops; every workgroup ends with a K*E partial that the second pass reads
  int nblk = (int)ctr_ceil_div(batch, 128);
  if (nblk > 512) nblk = 512;
  if ((int64_t)nblk * slab > workspace_floats) return CTR_OK;  // leave the bags to the generic kernel
  // bags sorted into launches by register tile: <= 2, <= 8, <= 32 rows
  const int caps[3] = {2, 8, 32};
  Bags sorted;
  sorted.n = 0;
  int start[4] = {0, 0, 0, 0};
  for (int c = 0; c < 3; ++c) {
    for (int k = 0; k < B.n; ++k) {
      const int lo = c == 0 ? 0 : caps[c - 1];
      if (B.b[k].rows > lo && B.b[k].rows <= caps[c]) sorted.b[sorted.n++] = B.b[k];
    }
    start[c + 1] = sorted.n;
  }
  for (int c = 0; c < 3; ++c) {
    const int cnt = start[c + 1] - start[c];
    if (cnt == 0) continue;
    int zmax = 1;
    for (int k = start[c]; k < start[c + 1]; ++k) zmax = (sorted.b[k].width + 63) / 64 > zmax ? (sorted.b[k].width + 63) / 64 : zmax;
    const dim3 grid(nblk, cnt, zmax);
    if (c == 0)
      hipLaunchKernelGGL(bag_bwd_kernel<2>, grid, dim3(kBlock), 0, st, sorted, start[c], ldx, (uint32_t)batch, gout, ldo,
                         workspace, slab);
    else if (c == 1)
      hipLaunchKernelGGL(bag_bwd_kernel<8>, grid, dim3(kBlock), 0, st, sorted, start[c], ldx, (uint32_t)batch, gout, ldo,
                         workspace, slab);
    else
      hipLaunchKernelGGL(bag_bwd_kernel<32>, grid, dim3(kBlock), 0, st, sorted, start[c], ldx, (uint32_t)batch, gout,
                         ldo, workspace, slab);
  }
  int rc = ctr_launch_status();
  if (rc != CTR_OK) return rc;
  CtrSegments segs;
  segs.n = 0;
  for (int k = 0; k < B.n; ++k) {
    const ctr_field_t& f = fields[idx_of[k]];
    segs.s[segs.n++] = CtrSegment{B.b[k].slab_off, (int64_t)f.bag_size * f.width, f.grad};
    handled[idx_of[k]] = 1;
  }
  *used_floats = (int64_t)nblk * slab;
  return ctr_reduce_segments(workspace, nblk, slab, segs, st);
}
