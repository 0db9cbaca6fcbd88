// Embedding stage: fused row gather (K1/K3) + weighted bag pooling (K2) + concat,
// and its backward (K11).  HBM-bound byte moving: no MFMA here by design.
//
// Work decomposition: the output row of one sample is cut into "units" of VEC
// floats (VEC = 4 -> one dwordx4 per lane).  Consecutive lanes take consecutive
// units of the flat (sample, unit) space, so every wave-instruction stores 1 KiB
// contiguous and reads its indices coalesced, whatever the field widths are; a
// 64-byte table row (E = 16) is fetched by 4 neighbouring lanes, i.e. one
// wave-instruction has 16 independent rows in flight.
#include "ctr_common.h"

#include <stdlib.h>

namespace {

constexpr int kMaxUnits = 4096;   // units per sample the per-block LUT can hold
constexpr int kBlock = 256;

struct FieldPack {
  ctr_field_t f[CTR_MAX_FIELDS];
};

struct UnitLut {
  ctr_field_t f[CTR_MAX_FIELDS];
  int start[CTR_MAX_FIELDS + 1];       // first unit of field
  unsigned short field_of[kMaxUnits];  // unit -> field
};

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
  using T = float;
  static __device__ __forceinline__ T zero() { return 0.0f; }
};
template <>
struct Vec<4> {
  using T = float4;
  static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
};

__device__ __forceinline__ float vfma(float w, float t, float a) { return fmaf(w, t, a); }
__device__ __forceinline__ float4 vfma(float w, float4 t, float4 a) {
  return make_float4(fmaf(w, t.x, a.x), fmaf(w, t.y, a.y), fmaf(w, t.z, a.z), fmaf(w, t.w, a.w));
}
__device__ __forceinline__ float vmul(float a, float b) { return a * b; }
__device__ __forceinline__ float4 vmul(float4 a, float4 b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float vscale(float w, float a) { return w * a; }
__device__ __forceinline__ float4 vscale(float w, float4 a) { return make_float4(w * a.x, w * a.y, w * a.z, w * a.w); }

__device__ __forceinline__ void atomic_add_vec(float* p, float v) { ctr_atomic_add_global(p, v); }
__device__ __forceinline__ void atomic_add_vec(float* p, float4 v) {
  ctr_atomic_add_global(p + 0, v.x);
  ctr_atomic_add_global(p + 1, v.y);
  ctr_atomic_add_global(p + 2, v.z);
  ctr_atomic_add_global(p + 3, v.w);
}

template <int VEC>
__device__ __forceinline__ void build_lut(UnitLut& s, const FieldPack& P, int nfields) {
  // descriptors: kernarg -> LDS, one dword per lane
  const int words = nfields * (int)(sizeof(ctr_field_t) / 4);
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&P);
  uint32_t* dst = reinterpret_cast<uint32_t*>(s.f);
  for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int f = 0; f < nfields; ++f) {
      s.start[f] = acc;
      acc += s.f[f].width / VEC;
    }
    s.start[nfields] = acc;
  }
  __syncthreads();
  const int upr = s.start[nfields];
  for (int u = threadIdx.x; u < upr; u += blockDim.x) {
    int f = 0;
    while (u >= s.start[f + 1]) ++f;
    s.field_of[u] = (unsigned short)f;
  }
  __syncthreads();
}

// row index of an id field for sample b; out-of-range -> row 0 + flag
__device__ __forceinline__ int64_t checked_row(int64_t r, int64_t vocab, int32_t* err_flag) {
  if (r < 0 || r >= vocab) {
    if (err_flag) *err_flag = 1;
    return 0;
  }
  return r;
}

template <int VEC>
__global__ void __launch_bounds__(kBlock)
embed_fwd_kernel(const FieldPack P, int nfields, const float* __restrict__ x, int64_t ldx, uint32_t batch,
                 float* __restrict__ out, int64_t ldo, int32_t* err_flag, const CtrFastDiv div) {
  using V = typename Vec<VEC>::T;
  __shared__ UnitLut s;
  build_lut<VEC>(s, P, nfields);
  const uint32_t upr = div.d;          // units per sample; batch * upr < 2^32 (checked on the host)
  const uint32_t total = batch * upr;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
    const uint32_t b = ctr_div(g, div);
    const uint32_t u = g - b * upr;
    const int fi = s.field_of[u];
    const ctr_field_t f = s.f[fi];  // by value: one batch of LDS reads, not a reload after every store
    const int off = (int)(u - (uint32_t)s.start[fi]) * VEC;
    V v;
    switch (f.kind) {
      case CTR_FIELD_ID_I64: {
        const int64_t r = checked_row(ctr_ldg(f.idx + (int64_t)b * f.idx_stride), f.vocab, err_flag);
        v = ctr_ldg(reinterpret_cast<const V*>(f.table + r * f.width + off));
      } break;
      case CTR_FIELD_ID_F32: {
        const int64_t r = checked_row((int64_t)x[(int64_t)b * ldx + f.src_col], f.vocab, err_flag);
        v = ctr_ldg(reinterpret_cast<const V*>(f.table + r * f.width + off));
      } break;
      case CTR_FIELD_BAG: {
        // in-order fp32 FMA chain over the K bag rows: a one-hot slice returns
        // the selected row bit-exactly, as the reference's matmul does
        const float* xr = x + (int64_t)b * ldx + f.src_col;
        v = Vec<VEC>::zero();
        for (int j = 0; j < f.bag_size; ++j)
          v = vfma(xr[j], ctr_ldg(reinterpret_cast<const V*>(f.table + (int64_t)j * f.width + off)), v);
      } break;
      case CTR_FIELD_DENSE: {
        v = *reinterpret_cast<const V*>(x + (int64_t)b * ldx + f.src_col + off);
      } break;
      default: {  // CTR_FIELD_PROD_I64
        const int64_t r1 = checked_row(ctr_ldg(f.idx + (int64_t)b * f.idx_stride), f.vocab, err_flag);
        const int64_t r2 = checked_row(ctr_ldg(f.idx2 + (int64_t)b * f.idx_stride), f.vocab2, err_flag);
        v = vmul(ctr_ldg(reinterpret_cast<const V*>(f.table + r1 * f.width + off)),
                 ctr_ldg(reinterpret_cast<const V*>(f.table2 + r2 * f.width + off)));
      } break;
    }
    *reinterpret_cast<V*>(out + (int64_t)b * ldo + f.out_col + off) = v;
  }
}

// Backward.  Id fields: one fp32 atomic per gradient element straight into the
// dense (V,E) gradient (rows are spread over a large table, the good case for
// memory-side atomics).  Bag fields: every sample hits the same <= 21 rows, the
// worst case for global atomics, so each block first reduces into an LDS copy of
// the bag tables' gradients (ds_add_f32) and flushes it once.
template <int VEC>
__global__ void __launch_bounds__(kBlock)
embed_bwd_kernel(const FieldPack P, int nfields, const float* __restrict__ x, int64_t ldx, uint32_t batch,
                 const float* __restrict__ gout, int64_t ldo, int bag_floats, int hot_floats, const CtrFastDiv div,
                 float* __restrict__ ws /* [gridDim.x][bag_floats] partials, or NULL -> atomics */) {
  using V = typename Vec<VEC>::T;
  __shared__ UnitLut s;
  __shared__ int s_bag_off[CTR_MAX_FIELDS];
  __shared__ int s_hot_off[CTR_MAX_FIELDS];
  extern __shared__ float s_bag[];  // bag_floats accumulators, then hot_floats for row 0 of the id fields
  float* s_hot = s_bag + bag_floats;
  build_lut<VEC>(s, P, nfields);
  if (threadIdx.x == 0) {
    int acc = 0, hot = 0;
    for (int f = 0; f < nfields; ++f) {
      s_bag_off[f] = acc;
      s_hot_off[f] = hot;
      if (s.f[f].kind == CTR_FIELD_BAG && s.f[f].grad) acc += s.f[f].bag_size * s.f[f].width;
      if ((s.f[f].kind == CTR_FIELD_ID_I64 || s.f[f].kind == CTR_FIELD_ID_F32) && s.f[f].grad) hot += s.f[f].width;
    }
  }
  for (int i = threadIdx.x; i < bag_floats + hot_floats; i += blockDim.x) s_bag[i] = 0.0f;
  __syncthreads();

  const uint32_t upr = div.d;
  const uint32_t total = batch * upr;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
    const uint32_t b = ctr_div(g, div);
    const uint32_t u = g - b * upr;
    const int fi = s.field_of[u];
    const ctr_field_t f = s.f[fi];  // by value (see the forward kernel)
    if (f.kind == CTR_FIELD_DENSE) continue;
    const int off = (int)(u - (uint32_t)s.start[fi]) * VEC;
    const V gv = *reinterpret_cast<const V*>(gout + (int64_t)b * ldo + f.out_col + off);
    switch (f.kind) {
      case CTR_FIELD_ID_I64:
      case CTR_FIELD_ID_F32: {
        if (!f.grad) break;
        int64_t r = f.kind == CTR_FIELD_ID_I64 ? ctr_ldg(f.idx + (int64_t)b * f.idx_stride)
                                                : (int64_t)x[(int64_t)b * ldx + f.src_col];
        if (r < 0 || r >= f.vocab) break;  // bad id: flagged by the forward, contributes to no row
        // Row 0 is the padding id of the behaviour sequences (scripts/din.py:23-31: a quarter of
        // a history is zeros): hundreds of thousands of same-address atomics would serialise
        // (~60 ns each).  It is summed per workgroup in LDS and flushed once.
        if (VEC == 1 && r == 0 && hot_floats > 0) atomicAdd(s_hot + s_hot_off[fi] + off, reinterpret_cast<const float*>(&gv)[0]);
        else atomic_add_vec(f.grad + r * f.width + off, gv);
      } break;
      case CTR_FIELD_BAG: {
        if (!f.grad) break;
        const float* xr = x + (int64_t)b * ldx + f.src_col;
        float* acc = s_bag + s_bag_off[fi] + off;
        for (int j = 0; j < f.bag_size; ++j) {
          const float w = xr[j];
          if (w != 0.0f) {
            const V c = vscale(w, gv);
            const float* cp = reinterpret_cast<const float*>(&c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) atomicAdd(acc + (int64_t)j * f.width + e, cp[e]);
          }
        }
      } break;
      default: {  // PROD: d(t1*t2) = g*t2, g*t1
        int64_t r1 = ctr_ldg(f.idx + (int64_t)b * f.idx_stride);
        int64_t r2 = ctr_ldg(f.idx2 + (int64_t)b * f.idx_stride);
        if (r1 < 0 || r1 >= f.vocab || r2 < 0 || r2 >= f.vocab2) break;  // bad id: no gradient
        const V t1 = ctr_ldg(reinterpret_cast<const V*>(f.table + r1 * f.width + off));
        const V t2 = ctr_ldg(reinterpret_cast<const V*>(f.table2 + r2 * f.width + off));
        if (f.grad) atomic_add_vec(f.grad + r1 * f.width + off, vmul(gv, t2));
        if (f.grad2) atomic_add_vec(f.grad2 + r2 * f.width + off, vmul(gv, t1));
      } break;
    }
  }
  __syncthreads();
  if (hot_floats > 0) {
    for (int fi = 0; fi < nfields; ++fi) {
      const ctr_field_t& f = s.f[fi];
      if ((f.kind != CTR_FIELD_ID_I64 && f.kind != CTR_FIELD_ID_F32) || !f.grad) continue;
      for (int i = threadIdx.x; i < f.width; i += blockDim.x) {
        const float v = s_hot[s_hot_off[fi] + i];
        if (v != 0.0f) ctr_atomic_add_global(f.grad + i, v);
      }
    }
  }
  if (ws) {
    // plain stores of this workgroup's partial; reduce.hip adds the partials up
    for (int i = threadIdx.x; i < bag_floats; i += blockDim.x) ws[(int64_t)blockIdx.x * bag_floats + i] = s_bag[i];
    return;
  }
  for (int fi = 0; fi < nfields; ++fi) {
    const ctr_field_t& f = s.f[fi];
    if (f.kind != CTR_FIELD_BAG || !f.grad) continue;
    const int n = f.bag_size * f.width;
    const float* acc = s_bag + s_bag_off[fi];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float v = acc[i];
      if (v != 0.0f) ctr_atomic_add_global(f.grad + i, v);
    }
  }
}

// ---------------------------------------------------------------------------
struct TablePtrs {
  const float* p[CTR_MAX_FIELDS];
  int64_t vocab[CTR_MAX_FIELDS];
};

template <int UNROLL>
__global__ void __launch_bounds__(kBlock)
embed_ids_fast_kernel(const TablePtrs T, int nfields, int lpr, int width, const int64_t* __restrict__ idx,
                      uint32_t items, float* __restrict__ out, int64_t ldo, int32_t* err_flag, const CtrFastDiv div) {
  __shared__ const float* s_tab[CTR_MAX_FIELDS];
  __shared__ int64_t s_vocab[CTR_MAX_FIELDS];
  if (threadIdx.x < nfields) {
    s_tab[threadIdx.x] = T.p[threadIdx.x];
    s_vocab[threadIdx.x] = T.vocab[threadIdx.x];
  }
  __syncthreads();
  const int sub = threadIdx.x % lpr;                       // which dwordx4 of the row
  const uint32_t per_block = (kBlock / lpr) * UNROLL;      // items per workgroup pass
  const uint32_t slot = threadIdx.x / lpr;
  for (uint32_t base = blockIdx.x * per_block; base < items; base += gridDim.x * per_block) {
    const float* src[UNROLL];
    float* dst[UNROLL];
    bool live[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      const uint32_t i = base + slot + k * (kBlock / lpr);
      live[k] = i < items;
      const uint32_t ii = live[k] ? i : 0;
      const uint32_t b = ctr_div(ii, div);
      const uint32_t f = ii - b * div.d;
      int64_t r = ctr_ldg(idx + ii);
      if (r < 0 || r >= s_vocab[f]) {
        if (err_flag) *err_flag = 1;
        r = 0;
      }
      src[k] = s_tab[f] + r * width + sub * 4;
      dst[k] = out + (int64_t)b * ldo + f * width + sub * 4;
    }
    ctr_f32x4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
#ifdef CTR_NT_LOADS
      v[k] = __builtin_nontemporal_load((const CTR_GLOBAL ctr_f32x4*)(src[k]));
#else
      v[k] = *(const CTR_GLOBAL ctr_f32x4*)(src[k]);
#endif
    }
#pragma unroll
    for (int k = 0; k < UNROLL; ++k)
      if (live[k]) {
#ifdef CTR_NT_STORES
        __builtin_nontemporal_store(v[k], (CTR_GLOBAL ctr_f32x4*)(dst[k]));
#else
        *(CTR_GLOBAL ctr_f32x4*)(dst[k]) = v[k];
#endif
      }
  }
}

// returns true (and launches) when the descriptor list matches the fast-path shape
bool try_fast_ids(const ctr_field_t* f, int n, int64_t batch, float* out, int64_t ldo, int32_t* err_flag,
                  hipStream_t st, int* rc) {
  const int w = f[0].width;
  if (w % 4 != 0 || w > 256) return false;
  const int lpr = w / 4;
  if ((lpr & (lpr - 1)) != 0 || lpr > 64) return false;
  if (!ctr_aligned16(out) || ldo % 4 != 0 || batch * n >= (1ll << 32)) return false;
  TablePtrs T;
  for (int i = 0; i < n; ++i) {
    if (f[i].kind != CTR_FIELD_ID_I64 || f[i].width != w || f[i].out_col != i * w) return false;
    if (f[i].idx != f[0].idx + i || f[i].idx_stride != n || !ctr_aligned16(f[i].table)) return false;
    T.p[i] = f[i].table;
    T.vocab[i] = f[i].vocab;
  }
#ifndef CTR_FAST_UNROLL
#define CTR_FAST_UNROLL 2  // A/B on MI355X (dev/gather_ab.py): 1/2/4/8 x grid caps are within 2 %, 2 x 8192 best on Zipf ids
#endif
  constexpr int kUnroll = CTR_FAST_UNROLL;
  const uint32_t items = (uint32_t)(batch * n);
  const uint32_t per_block = (kBlock / lpr) * kUnroll;
  int64_t grid = ctr_ceil_div(items, per_block);
#ifndef CTR_FAST_GRID
#define CTR_FAST_GRID (256 * 32)
#endif
  if (grid > CTR_FAST_GRID) grid = CTR_FAST_GRID;
  hipLaunchKernelGGL(embed_ids_fast_kernel<kUnroll>, dim3((unsigned)grid), dim3(kBlock), 0, st, T, n, lpr, w, f[0].idx,
                     items, out, ldo, err_flag, ctr_fastdiv((uint32_t)n));
  *rc = ctr_launch_status();
  return true;
}

// Row fields of ONE width whose ids come from arbitrary int64 columns, products of two rows included (NeuralCF:
// model/neuralcf.py:35-38, two MLP rows and the GMF product per sample).  Same loop as above -- lpr lanes move one
// dwordx4 each of a (sample, field) item, UNROLL items in flight -- with the field's pointers read from a small
// LDS table.  The generic kernel copies a 96-byte descriptor out of LDS per 16 bytes it moves: 18.3 us on
// 65536 x 3 x 64 against the 8 us its 50 MB of writes need.
struct RowField {
  const int64_t* idx; const int64_t* idx2;     // idx2 != NULL: product of two rows
  const float* table; const float* table2;
  int64_t stride, vocab, vocab2;
  int out_col, pad;
};
struct RowFields {
  RowField f[8];
};

template <int UNROLL>
__global__ void __launch_bounds__(kBlock)
embed_rows_fast_kernel(const RowFields F, int nfields, int lpr, int width, uint32_t items, float* __restrict__ out,
                       int64_t ldo, int32_t* err_flag, const CtrFastDiv div) {
  __shared__ RowField s_f[8];
  for (int i = threadIdx.x; i < nfields * (int)(sizeof(RowField) / 4); i += blockDim.x)
    reinterpret_cast<uint32_t*>(s_f)[i] = reinterpret_cast<const uint32_t*>(F.f)[i];
  __syncthreads();
  const int sub = threadIdx.x % lpr;
  const uint32_t per_block = (kBlock / lpr) * UNROLL;
  const uint32_t slot = threadIdx.x / lpr;
  for (uint32_t base = blockIdx.x * per_block; base < items; base += gridDim.x * per_block) {
    const float* src[UNROLL];
    const float* src2[UNROLL];
    float* dst[UNROLL];
    bool live[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      const uint32_t i = base + slot + k * (kBlock / lpr);
      live[k] = i < items;
      const uint32_t ii = live[k] ? i : 0;
      const uint32_t b = ctr_div(ii, div);
      const uint32_t f = ii - b * div.d;
      const int64_t* ip = s_f[f].idx;
      const int64_t* ip2 = s_f[f].idx2;
      const int64_t stride = s_f[f].stride;
      int64_t r = ctr_ldg(ip + (int64_t)b * stride);
      int64_t r2 = ip2 ? ctr_ldg(ip2 + (int64_t)b * stride) : 0;
      if (r < 0 || r >= s_f[f].vocab || (ip2 && (r2 < 0 || r2 >= s_f[f].vocab2))) {
        if (err_flag) *err_flag = 1;
        if (r < 0 || r >= s_f[f].vocab) r = 0;
        if (ip2 && (r2 < 0 || r2 >= s_f[f].vocab2)) r2 = 0;
      }
      src[k] = s_f[f].table + r * width + sub * 4;
      src2[k] = ip2 ? s_f[f].table2 + r2 * width + sub * 4 : nullptr;
      dst[k] = out + (int64_t)b * ldo + s_f[f].out_col + sub * 4;
    }
    ctr_f32x4 v[UNROLL], v2[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      v[k] = *(const CTR_GLOBAL ctr_f32x4*)(src[k]);
      v2[k] = ctr_f32x4{1.0f, 1.0f, 1.0f, 1.0f};
      if (src2[k]) v2[k] = *(const CTR_GLOBAL ctr_f32x4*)(src2[k]);
    }
#pragma unroll
    for (int k = 0; k < UNROLL; ++k)
      if (live[k]) *(CTR_GLOBAL ctr_f32x4*)(dst[k]) = src2[k] ? v[k] * v2[k] : v[k];
  }
}

bool try_fast_rows(const ctr_field_t* f, int n, int64_t batch, float* out, int64_t ldo, int32_t* err_flag,
                   hipStream_t st, int* rc) {
  if (n < 1 || n > 8) return false;
  const int w = f[0].width;
  if (w % 4 != 0 || w > 256) return false;
  const int lpr = w / 4;
  if ((lpr & (lpr - 1)) != 0 || lpr > 64) return false;
  if (!ctr_aligned16(out) || ldo % 4 != 0 || batch * n >= (1ll << 32)) return false;
  RowFields F;
  for (int i = 0; i < n; ++i) {
    const bool prod = f[i].kind == CTR_FIELD_PROD_I64;
    if (f[i].kind != CTR_FIELD_ID_I64 && !prod) return false;
    if (f[i].width != w || f[i].out_col % 4 != 0 || !ctr_aligned16(f[i].table) || (prod && !ctr_aligned16(f[i].table2)))
      return false;
    F.f[i] = RowField{f[i].idx, prod ? f[i].idx2 : nullptr, f[i].table, prod ? f[i].table2 : nullptr,
                      f[i].idx_stride, f[i].vocab, prod ? f[i].vocab2 : 0, f[i].out_col, 0};
  }
  constexpr int kUnroll = 4;
  const uint32_t items = (uint32_t)(batch * n);
  const uint32_t per_block = (kBlock / lpr) * kUnroll;
  int64_t grid = ctr_ceil_div(items, per_block);
  if (grid > 256 * 8) grid = 256 * 8;   // A/B on NeuralCF's stage (unroll x cap): 4x8 17.1 us, 4x16 18.2, 2x16 17.7, 8x16 21.8
  hipLaunchKernelGGL(embed_rows_fast_kernel<kUnroll>, dim3((unsigned)grid), dim3(kBlock), 0, st, F, n, lpr, w, items, out,
                     ldo, err_flag, ctr_fastdiv((uint32_t)n));
  *rc = ctr_launch_status();
  return true;
}

// Backward of the same shape (F id fields of one width out of one (B,F) index matrix): the flat
// element index g IS the offset into a dense gout, one dword per lane, so a wave-instruction adds to
// 256/(4*width) whole gradient rows.  No descriptor LUT, no per-element struct copy: the generic
// kernel spends ~20 LDS reads per element on those and runs at 0.4x the atomic rate on this shape
// (239 us vs 94 us at 26 x 1e6 x 16, batch 65536; dev/gather_ceiling.hip has the bare loop).
struct GradPtrs {
  float* p[CTR_MAX_FIELDS];
  int64_t vocab[CTR_MAX_FIELDS];
};

__global__ void __launch_bounds__(kBlock)
embed_ids_fast_bwd_kernel(const GradPtrs G, int nfields, int wshift, const int64_t* __restrict__ idx, uint32_t items,
                          const float* __restrict__ gout, int64_t ldo, const CtrFastDiv div) {
  __shared__ float* s_grad[CTR_MAX_FIELDS];
  __shared__ int64_t s_vocab[CTR_MAX_FIELDS];
  extern __shared__ float s_row0[];  // [nfields][width]: row 0 (the sequences' padding id) summed per workgroup
  if (threadIdx.x < nfields) {
    s_grad[threadIdx.x] = G.p[threadIdx.x];
    s_vocab[threadIdx.x] = G.vocab[threadIdx.x];
  }
  for (int i = threadIdx.x; i < (nfields << wshift); i += blockDim.x) s_row0[i] = 0.0f;
  __syncthreads();
  const uint32_t width = 1u << wshift;
  const uint32_t total = items << wshift;  // < 2^32 (checked on the host)
  bool any0 = false;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
    const uint32_t item = g >> wshift, e = g & (width - 1);
    const uint32_t b = ctr_div(item, div);
    const uint32_t f = item - b * div.d;
    const int64_t r = ctr_ldg(idx + item);
    const float v = ctr_ldg(gout + (int64_t)b * ldo + (f << wshift) + e);
    float* dst = s_grad[f];
    if (dst == nullptr || r < 0 || r >= s_vocab[f]) continue;  // frozen table / bad id: no gradient
    if (r == 0) {  // same-address global atomics serialise (~60 ns each): see embed_bwd_kernel
      atomicAdd(s_row0 + (f << wshift) + e, v);
      any0 = true;
    } else {
      ctr_atomic_add_global(dst + (r << wshift) + e, v);
    }
  }
  if (__syncthreads_or(any0)) {
    for (int i = threadIdx.x; i < (nfields << wshift); i += blockDim.x) {
      const float v = s_row0[i];
      float* dst = s_grad[i >> wshift];
      if (v != 0.0f && dst) ctr_atomic_add_global(dst + (i & (width - 1)), v);
    }
  }
}

// The same scatter with the rows a workgroup hits REPEATEDLY summed in LDS first (the backward half of "LDS staging of
// hot rows").  fp32 atomics on one address are serialised by the memory side (~25-30 ns each, and they block the
// requests queued behind them): under a Zipf law the hottest row of a 65536-sample batch is hit ~1900 times per field
// and the launch took 170 us against 110 us for uniform ids.  Which rows are hot is not known in advance, so every
// workgroup finds out for its own slice: a direct-mapped LDS cache of kHotSlots accumulator rows with admission on the
// SECOND touch -- `seen[s]` remembers the last (field, row) that hashed to slot s and is overwritten freely; an item
// that finds its own key there claims `tag[s]` (compare-and-swap from empty, never replaced) and from then on every
// item of that row adds into the slot's LDS accumulator instead of issuing a global atomic; the accumulators are
// flushed with one atomic row per slot at the end.  Rows seen once (uniform ids: practically all of them) take the
// unchanged global path after two LDS reads and a write.  Workgroups walk CONTIGUOUS slices (so that the repeats of
// a row meet in one cache) and there are fewer of them than in the plain kernel.
constexpr uint32_t kHotEmpty = 0xffffffffu;

template <int SLOTS_LOG2>
__global__ void __launch_bounds__(1024)
embed_ids_hot_bwd_kernel(const GradPtrs G, int nfields, int wshift, const int64_t* __restrict__ idx, uint32_t items,
                         const float* __restrict__ gout, int64_t ldo, const CtrFastDiv div, uint32_t items_per_block) {
  __shared__ float* s_grad[CTR_MAX_FIELDS];
  __shared__ int64_t s_vocab[CTR_MAX_FIELDS];
  constexpr int kHotSlots = 1 << SLOTS_LOG2;
  __shared__ uint32_t s_seen[kHotSlots], s_tag[kHotSlots];
  extern __shared__ float s_acc[];  // [kHotSlots][width]
  const uint32_t width = 1u << wshift;
  if (threadIdx.x < nfields) {
    s_grad[threadIdx.x] = G.p[threadIdx.x];
    s_vocab[threadIdx.x] = G.vocab[threadIdx.x];
  }
  for (int i = threadIdx.x; i < kHotSlots; i += blockDim.x) s_seen[i] = s_tag[i] = kHotEmpty;
  for (int i = threadIdx.x; i < (kHotSlots << wshift); i += blockDim.x) s_acc[i] = 0.0f;
  __syncthreads();
  const uint32_t item0 = blockIdx.x * items_per_block;
  const uint32_t item1 = item0 + items_per_block < items ? item0 + items_per_block : items;
  const uint32_t g0 = item0 << wshift, g1 = item1 << wshift;   // (items << wshift < 2^32: checked on the host)
  // kHotUnroll elements per thread and round: every load of a round is requested before any is used (two workgroups
  // per CU fit beside the 64 KB of accumulators, so a round trip per element would be the whole kernel)
  constexpr int kHotUnroll = 8;
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t gb = g0 + threadIdx.x; gb < g1; gb += blockDim.x * kHotUnroll) {
    int64_t r[kHotUnroll];
    float v[kHotUnroll];
    uint32_t f[kHotUnroll];
#pragma unroll
    for (int k = 0; k < kHotUnroll; ++k) {
      uint32_t g = gb + k * blockDim.x;
      g = g < g1 ? g : g1 - 1;                     // (clamped: loaded, not used)
      const uint32_t item = g >> wshift, e = g & (width - 1);
      const uint32_t b = ctr_div(item, div);
      f[k] = item - b * div.d;
      r[k] = ctr_ldg(idx + item);
      v[k] = ctr_ldg(gout + (int64_t)b * ldo + (f[k] << wshift) + e);
    }
#pragma unroll
    for (int k = 0; k < kHotUnroll; ++k) {
      const uint32_t g = gb + k * blockDim.x;
      const uint32_t e = g & (width - 1);
      float* dst = s_grad[f[k]];
      const bool ok = g < g1 && dst != nullptr && r[k] >= 0 && r[k] < s_vocab[f[k]];   // frozen table / bad id: no gradient
      const uint32_t key = (f[k] << 24) | (uint32_t)r[k];      // (every vocabulary < 2^24: checked on the host)
      const uint32_t slot = (key * 2654435761u) >> (32 - SLOTS_LOG2);
      // the first lane of an item (its `width` lanes are consecutive and share key) looks the row up for all of them
      int hot = 0;
      if (ok && e == 0) {
        hot = s_tag[slot] == key;
        if (!hot) {
          if (s_seen[slot] == key) {
            // second touch in this workgroup: claim the slot if nobody owns it
            const uint32_t old = atomicCAS(&s_tag[slot], kHotEmpty, key);
            hot = old == kHotEmpty || old == key;
          } else {
            s_seen[slot] = key;
          }
        }
      }
      // (width 16: an item is one DPP row, its first lane's answer is a row broadcast)
      hot = wshift == 4 ? __builtin_amdgcn_update_dpp(0, hot, 0x150, 0xf, 0xf, false)
                        : __shfl(hot, (int)(lane & ~(width - 1)), 64);
      if (ok) {
        if (hot) atomicAdd(s_acc + (slot << wshift) + e, v[k]);
        else ctr_atomic_add_global(dst + (r[k] << wshift) + e, v[k]);
      }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < ((uint32_t)kHotSlots << wshift); i += blockDim.x) {
    const uint32_t key = s_tag[i >> wshift];
    if (key == kHotEmpty) continue;
    const float v = s_acc[i];
    float* dst = s_grad[key >> 24];
    ctr_atomic_add_global(dst + ((int64_t)(key & 0xffffffu) << wshift) + (i & (width - 1)), v);
  }
}

bool try_fast_ids_bwd(const ctr_field_t* f, int n, int64_t batch, const float* gout, int64_t ldo, hipStream_t st,
                      int* rc) {
  const int w = f[0].width;
  if (w <= 0 || w > 64 || (w & (w - 1)) != 0) return false;  // wider rows: the generic kernel's shape is fine
  if (batch * n * w >= (1ll << 32)) return false;
  GradPtrs G;
  int wshift = 0;
  while ((1 << wshift) < w) ++wshift;
  for (int i = 0; i < n; ++i) {
    if (f[i].kind != CTR_FIELD_ID_I64 || f[i].width != w || f[i].out_col != i * w) return false;
    if (f[i].idx != f[0].idx + i || f[i].idx_stride != n) return false;
    G.p[i] = f[i].grad;
    G.vocab[i] = f[i].vocab;
  }
  const uint32_t items = (uint32_t)(batch * n);
  // large stages: per-workgroup pre-reduction of repeated rows (embed_ids_hot_bwd_kernel); CTR_EMBED_HOT=0 keeps the plain
  // scatter for A/B
  static const int hot_mode = [] { const char* e = getenv("CTR_EMBED_HOT"); return e ? atoi(e) : 1; }();
  bool small_vocab = true;
  for (int i = 0; i < n; ++i) small_vocab = small_vocab && f[i].vocab < (1 << 24);
  if (hot_mode && small_vocab && n <= 255 && w <= 32 && items >= 512u * 1024u) {
    static const int hot_wgs = [] { const char* e = getenv("CTR_EMBED_HOT_WGS"); return e ? atoi(e) : 256; }();
    static const int hot_slots = [] { const char* e = getenv("CTR_EMBED_HOT_SLOTS"); return e ? atoi(e) : 11; }();
    static const int hot_threads = [] { const char* e = getenv("CTR_EMBED_HOT_THREADS"); return e ? atoi(e) : 1024; }();
    const uint32_t per = (uint32_t)ctr_ceil_div(items, hot_wgs);
    const unsigned grid = (unsigned)ctr_ceil_div(items, per);
#define CTR_HOT_LAUNCH(L)                                                                                              \
  do {                                                                                                                 \
    const size_t lds = ((size_t)1 << L) * w * sizeof(float);                                                           \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(embed_ids_hot_bwd_kernel<L>),                                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {                     \
      *rc = CTR_ELAUNCH;                                                                                               \
      return true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL(embed_ids_hot_bwd_kernel<L>, dim3(grid), dim3(hot_threads), lds, st, G, n, wshift, f[0].idx, items,  \
                       gout, ldo, ctr_fastdiv((uint32_t)n), per);                                                      \
  } while (0)
    if (hot_slots == 9 || ((size_t)2048 * w * sizeof(float) > 140 * 1024 && hot_slots > 9)) CTR_HOT_LAUNCH(9);
    else if (hot_slots == 10) CTR_HOT_LAUNCH(10);
    else CTR_HOT_LAUNCH(11);
#undef CTR_HOT_LAUNCH
    *rc = ctr_launch_status();
    return true;
  }
  const int grid = ctr_stream_grid((int64_t)items * w, kBlock);
  hipLaunchKernelGGL(embed_ids_fast_bwd_kernel, dim3((unsigned)grid), dim3(kBlock), (size_t)n * w * sizeof(float), st, G, n,
                     wshift, f[0].idx, items, gout, ldo, ctr_fastdiv((uint32_t)n));
  *rc = ctr_launch_status();
  return true;
}

struct Plan {
  FieldPack pack;
  int vec;
  int units;
  int bag_floats;
};

int make_plan(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx, const float* io, int64_t ldo,
              bool backward, Plan* plan) {
  CTR_REQUIRE(fields && nfields > 0 && nfields <= CTR_MAX_FIELDS && io, CTR_EINVAL);
  bool vec4 = ctr_aligned16(io) && (ldo % 4 == 0);
  int floats = 0, bag_floats = 0;
  bool needs_x = false;
  for (int i = 0; i < nfields; ++i) {
    const ctr_field_t& f = fields[i];
    CTR_REQUIRE(f.kind >= CTR_FIELD_ID_I64 && f.kind <= CTR_FIELD_PROD_I64, CTR_EINVAL);
    CTR_REQUIRE(f.width > 0 && f.out_col >= 0 && (int64_t)f.out_col + f.width <= ldo, CTR_EINVAL);
    if (f.kind != CTR_FIELD_DENSE) {
      CTR_REQUIRE(f.table && f.vocab > 0, CTR_EINVAL);
      vec4 = vec4 && ctr_aligned16(f.table);
    }
    if (f.kind == CTR_FIELD_ID_I64 || f.kind == CTR_FIELD_PROD_I64) CTR_REQUIRE(f.idx && f.idx_stride >= 0, CTR_EINVAL);
    if (f.kind == CTR_FIELD_PROD_I64) {
      CTR_REQUIRE(f.idx2 && f.table2 && f.vocab2 > 0, CTR_EINVAL);
      vec4 = vec4 && ctr_aligned16(f.table2);
    }
    if (f.kind == CTR_FIELD_ID_F32 || f.kind == CTR_FIELD_BAG || f.kind == CTR_FIELD_DENSE) {
      needs_x = true;
      CTR_REQUIRE(f.src_col >= 0, CTR_EINVAL);
    }
    if (f.kind == CTR_FIELD_BAG) {
      CTR_REQUIRE(f.bag_size > 0 && f.bag_size == f.vocab && f.src_col + f.bag_size <= ldx, CTR_EINVAL);
      if (backward && f.grad) bag_floats += f.bag_size * f.width;
    }
    if (f.kind == CTR_FIELD_DENSE) {
      CTR_REQUIRE(f.src_col + f.width <= ldx, CTR_EINVAL);
      vec4 = vec4 && ctr_aligned16(x) && (ldx % 4 == 0) && (f.src_col % 4 == 0);
    }
    if (backward) {
      if (f.grad) vec4 = vec4 && ctr_aligned16(f.grad);
      if (f.kind == CTR_FIELD_PROD_I64 && f.grad2) vec4 = vec4 && ctr_aligned16(f.grad2);
    }
    vec4 = vec4 && (f.width % 4 == 0) && (f.out_col % 4 == 0);
    floats += f.width;
    plan->pack.f[i] = f;
  }
  if (needs_x) CTR_REQUIRE(x && ldx > 0, CTR_EINVAL);
  // backward: one dword per lane so that a wave's atomic instruction adds to 256
  // contiguous bytes of a gradient row (the shape that runs at the full memory-side
  // atomic rate; dwordx4-per-lane would stride the lanes 16 B apart)
  plan->vec = (vec4 && !backward) ? 4 : 1;
  plan->units = floats / plan->vec;
  plan->bag_floats = bag_floats;
  CTR_REQUIRE(plan->units <= kMaxUnits, CTR_ELIMIT);
  CTR_REQUIRE(bag_floats * 4 <= 96 * 1024, CTR_ELIMIT);
  return CTR_OK;
}

}  // namespace

extern "C" int ctr_embed_fwd(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx, int64_t batch,
                             float* out, int64_t ldo, int32_t* err_flag, void* stream) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;  // empty batch: nothing to read, pointers may be null
  CTR_REQUIRE(ldo > 0, CTR_EINVAL);
  Plan plan;
  int rc = make_plan(fields, nfields, x, ldx, out, ldo, false, &plan);
  if (rc != CTR_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (try_fast_ids(fields, nfields, batch, out, ldo, err_flag, st, &rc)) return rc;
  if (try_fast_rows(fields, nfields, batch, out, ldo, err_flag, st, &rc)) return rc;
  // wide bag fields run in a kernel of their own (embed_bag.hip); the gather kernel takes the rest
  unsigned char handled[CTR_MAX_FIELDS];
  rc = ctr_embed_fwd_bags(fields, nfields, x, ldx, batch, out, ldo, handled, st);
  if (rc != CTR_OK) return rc;
  ctr_field_t rest[CTR_MAX_FIELDS];
  int nrest = 0;
  for (int i = 0; i < nfields; ++i)
    if (!handled[i]) rest[nrest++] = fields[i];
  if (nrest == 0) return CTR_OK;
  if (nrest < nfields) {
    fields = rest;
    nfields = nrest;
    rc = make_plan(fields, nfields, x, ldx, out, ldo, false, &plan);
    if (rc != CTR_OK) return rc;
  }
  CTR_REQUIRE(batch * plan.units < (1ll << 32), CTR_ELIMIT);
  const CtrFastDiv div = ctr_fastdiv((uint32_t)plan.units);
  const int grid = ctr_stream_grid(batch * plan.units, kBlock);
  if (plan.vec == 4)
    hipLaunchKernelGGL(embed_fwd_kernel<4>, dim3(grid), dim3(kBlock), 0, st, plan.pack, nfields, x, ldx,
                       (uint32_t)batch, out, ldo, err_flag, div);
  else
    hipLaunchKernelGGL(embed_fwd_kernel<1>, dim3(grid), dim3(kBlock), 0, st, plan.pack, nfields, x, ldx,
                       (uint32_t)batch, out, ldo, err_flag, div);
  return ctr_launch_status();
}

static int embed_bwd_impl(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx, int64_t batch,
                          const float* gout, int64_t ldo, float* workspace, int64_t workspace_floats, void* stream) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(ldo > 0, CTR_EINVAL);
  Plan plan;
  int rc = make_plan(fields, nfields, x, ldx, gout, ldo, true, &plan);  // validates every descriptor
  if (rc != CTR_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  // small tables under a large batch: bucket the samples by row and reduce runs in
  // registers (embed_sorted.hip); what it takes is dropped from the scatter launch below
  unsigned char handled[CTR_MAX_FIELDS];
  int64_t sort_floats = 0, bag_floats_used = 0;
  rc = ctr_embed_bwd_sorted(fields, nfields, x, ldx, batch, gout, ldo, workspace, workspace_floats, &sort_floats,
                            handled, st);
  if (rc != CTR_OK) return rc;
  workspace_floats -= sort_floats;  // the sort buffers sit at the end
  {  // nothing taken by the sorted path and every field an id column of one (B,F) matrix: the lean scatter
    bool untouched = true;
    for (int i = 0; i < nfields; ++i) untouched = untouched && !handled[i];
    if (untouched && try_fast_ids_bwd(fields, nfields, batch, gout, ldo, st, &rc)) return rc;
  }
  // bag tables: register accumulation per output column (embed_bag.hip), partials at the start
  rc = ctr_embed_bwd_bags(fields, nfields, x, ldx, batch, gout, ldo, workspace, workspace_floats, &bag_floats_used,
                          handled, st);
  if (rc != CTR_OK) return rc;
  if (workspace) workspace += bag_floats_used;
  workspace_floats -= bag_floats_used;
  ctr_field_t rest[CTR_MAX_FIELDS];
  int nrest = 0;
  for (int i = 0; i < nfields; ++i) {
    const ctr_field_t& f = fields[i];
    const bool wants = f.kind != CTR_FIELD_DENSE && (f.grad || (f.kind == CTR_FIELD_PROD_I64 && f.grad2));
    if (!handled[i] && wants) rest[nrest++] = f;
  }
  if (nrest == 0) return CTR_OK;
  fields = rest;
  nfields = nrest;
  rc = make_plan(fields, nfields, x, ldx, gout, ldo, true, &plan);
  if (rc != CTR_OK) return rc;
  CTR_REQUIRE(batch * plan.units < (1ll << 32), CTR_ELIMIT);
  const CtrFastDiv div = ctr_fastdiv((uint32_t)plan.units);
  int grid = ctr_stream_grid(batch * plan.units, kBlock);
  // every workgroup ends with a partial of the bag tables' gradients: with a workspace
  // they are stored and summed by reduce.hip, otherwise added with (serialising) atomics
  const bool slabs = plan.bag_floats > 0 && workspace != nullptr;
  int cap = plan.bag_floats > 4096 ? 256 : 1024;
  if (slabs) {
    const int64_t fit = workspace_floats / plan.bag_floats;
    if (fit < cap) cap = (int)fit;
  } else if (plan.bag_floats > 0) {
    cap = plan.bag_floats > 4096 ? 64 : 128;
  }
  CTR_REQUIRE(cap >= 1, CTR_ELIMIT);
  if (grid > cap) grid = cap;
  int hot_floats = 0;
  for (int i = 0; i < nfields; ++i)
    if ((fields[i].kind == CTR_FIELD_ID_I64 || fields[i].kind == CTR_FIELD_ID_F32) && fields[i].grad)
      hot_floats += fields[i].width;
  if (hot_floats > 4096) hot_floats = 0;  // keep the LDS footprint small: very wide stages scatter row 0 directly
  const size_t dyn = (size_t)(plan.bag_floats + hot_floats) * sizeof(float);
  float* ws = slabs ? workspace : nullptr;
  if (plan.vec == 4)
    hipLaunchKernelGGL(embed_bwd_kernel<4>, dim3(grid), dim3(kBlock), dyn, st, plan.pack, nfields, x, ldx,
                       (uint32_t)batch, gout, ldo, plan.bag_floats, hot_floats, div, ws);
  else
    hipLaunchKernelGGL(embed_bwd_kernel<1>, dim3(grid), dim3(kBlock), dyn, st, plan.pack, nfields, x, ldx,
                       (uint32_t)batch, gout, ldo, plan.bag_floats, hot_floats, div, ws);
  rc = ctr_launch_status();
  if (rc != CTR_OK || !slabs) return rc;
  CtrSegments segs;
  segs.n = 0;
  int64_t off = 0;
  for (int i = 0; i < nfields; ++i) {
    const ctr_field_t& f = fields[i];
    if (f.kind == CTR_FIELD_BAG && f.grad) {
      const int64_t cnt = (int64_t)f.bag_size * f.width;
      segs.s[segs.n++] = CtrSegment{off, cnt, f.grad};
      off += cnt;
    }
  }
  return ctr_reduce_segments(workspace, grid, plan.bag_floats, segs, st);
}

extern "C" int ctr_embed_bwd(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx, int64_t batch,
                             const float* gout, int64_t ldo, float* workspace, int64_t workspace_floats,
                             void* stream) {
  return embed_bwd_impl(fields, nfields, x, ldx, batch, gout, ldo, workspace, workspace_floats, stream);
}
