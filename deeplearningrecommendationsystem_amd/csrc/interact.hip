// Feature-interaction kernels on the stacked embedding matrix emb (B, nvec*E) that
// the embedding stage wrote: PNN inner products, the DeepFM wide + FM second-order
// term, the FFM field-aware cross + logistic head, and the small elementwise
// activation backward the residual blocks need.  All are HBM-bound per-sample
// reductions: a workgroup stages a tile of samples in LDS (vectors padded so that
// lanes reading different vectors at the same offset hit different banks), reduces
// there, and writes coalesced results.  No MFMA by design.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kLdsFloats = 12288;  // 48 KiB of staged vectors per workgroup
constexpr int kMaxPairs = 64;      // explicit pair list (FFM uses 15)
constexpr int kMaxDense = 64;      // dense feature columns of the LR part (43 in the zoo)

struct Geometry {
  int nvec, e, vec, estride, row;  // row = floats per staged sample
  int tb;                          // samples per tile
};

inline Geometry make_geometry(int nvec, int e, bool aligned, int extra_per_sample = 0) {
  Geometry g;
  g.nvec = nvec;
  g.e = e;
  g.vec = (aligned && e % 4 == 0) ? 4 : 1;
  g.estride = g.vec == 4 ? e + 4 : e + 1;
  g.row = nvec * g.estride;
  int tb = kLdsFloats / (g.row + extra_per_sample);
  if (tb > 64) tb = 64;
  g.tb = tb;
  return g;
}

template <int VEC>
__device__ __forceinline__ void stage_vectors(float* lds, const Geometry& g, const float* __restrict__ emb, int64_t lde,
                                              int64_t b0, int count) {
  const int chunks = g.e / VEC;
  const int per_sample = g.nvec * chunks;
  for (int i = threadIdx.x; i < count * per_sample; i += blockDim.x) {
    const int s = i / per_sample, rem = i - s * per_sample;
    const int f = rem / chunks, c = rem - f * chunks;
    const float* src = emb + (b0 + s) * lde + f * g.e + c * VEC;
    float* dst = lds + s * g.row + f * g.estride + c * VEC;
    if (VEC == 4)
      *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
    else
      *dst = *src;
  }
}

template <int VEC>
__device__ __forceinline__ float dot_lds(const float* a, const float* b, int e) {
  float acc = 0.0f;
  if (VEC == 4) {
    for (int k = 0; k < e; k += 4) {
      const float4 x = *reinterpret_cast<const float4*>(a + k);
      const float4 y = *reinterpret_cast<const float4*>(b + k);
      acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
    }
  } else {
    for (int k = 0; k < e; ++k) acc = fmaf(a[k], b[k], acc);
  }
  return acc;
}

// ------------------------------------------------------------------ PNN inner
// p[b, idx(i,j)] = <v_i, v_j>, i < j lexicographic (model/pnn.py:59-66)
__device__ __forceinline__ int pair_index(int i, int j, int n) { return i * n - i * (i + 1) / 2 + (j - i - 1); }

template <int VEC>
__global__ void __launch_bounds__(kBlock)
allpairs_fwd_kernel(const Geometry g, const float* __restrict__ emb, int64_t lde, int64_t batch,
                    float* __restrict__ out, int64_t ldo) {
  extern __shared__ float lds[];
  const int npairs = g.nvec * (g.nvec - 1) / 2;
  for (int64_t b0 = (int64_t)blockIdx.x * g.tb; b0 < batch; b0 += (int64_t)gridDim.x * g.tb) {
    const int count = (int)(batch - b0 < g.tb ? batch - b0 : g.tb);
    stage_vectors<VEC>(lds, g, emb, lde, b0, count);
    __syncthreads();
    for (int w = threadIdx.x; w < count * npairs; w += blockDim.x) {
      const int s = w / npairs, p = w - s * npairs;
      // invert pair_index: walk the rows of the strict upper triangle
      int i = 0, rem = p;
      while (rem >= g.nvec - 1 - i) { rem -= g.nvec - 1 - i; ++i; }
      const int j = i + 1 + rem;
      const float* base = lds + s * g.row;
      out[(b0 + s) * ldo + p] = dot_lds<VEC>(base + i * g.estride, base + j * g.estride, g.e);
    }
    __syncthreads();
  }
}

// gemb[b, i, :] (+)= sum_{j != i} gp[b, idx(i,j)] * v_j
template <int VEC>
__global__ void __launch_bounds__(kBlock)
allpairs_bwd_kernel(const Geometry g, const float* __restrict__ emb, int64_t lde, int64_t batch,
                    const float* __restrict__ gp, int64_t ldgp, float* __restrict__ gemb, int64_t ldg,
                    int accumulate) {
  extern __shared__ float lds[];
  const int npairs = g.nvec * (g.nvec - 1) / 2;
  float* s_gp = lds + g.tb * g.row;  // [tb][npairs]
  const int chunks = g.e / VEC;
  for (int64_t b0 = (int64_t)blockIdx.x * g.tb; b0 < batch; b0 += (int64_t)gridDim.x * g.tb) {
    const int count = (int)(batch - b0 < g.tb ? batch - b0 : g.tb);
    stage_vectors<VEC>(lds, g, emb, lde, b0, count);
    for (int w = threadIdx.x; w < count * npairs; w += blockDim.x) {
      const int s = w / npairs, p = w - s * npairs;
      s_gp[s * npairs + p] = gp[(b0 + s) * ldgp + p];
    }
    __syncthreads();
    const int per_sample = g.nvec * chunks;
    for (int w = threadIdx.x; w < count * per_sample; w += blockDim.x) {
      const int s = w / per_sample, rem = w - s * per_sample;
      const int i = rem / chunks, c = (rem - i * chunks) * VEC;
      const float* base = lds + s * g.row;
      const float* gps = s_gp + s * npairs;
      float acc[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
      for (int j = 0; j < g.nvec; ++j) {
        if (j == i) continue;
        const float coef = gps[j > i ? pair_index(i, j, g.nvec) : pair_index(j, i, g.nvec)];
        const float* vj = base + j * g.estride + c;
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fmaf(coef, vj[v], acc[v]);
      }
      float* dst = gemb + (b0 + s) * ldg + i * g.e + c;
#pragma unroll
      for (int v = 0; v < VEC; ++v) dst[v] = accumulate ? dst[v] + acc[v] : acc[v];
    }
    __syncthreads();
  }
}

// ------------------------------------------------- logistic ("wide") part shared
struct WideArgs {
  const float* x;        // (B, ldx) feature matrix
  int64_t ldx;
  int user_col, item_col, dense_col0, ndense;
  const float* user1;    // (num_users, 1)
  const float* item1;    // (num_items, 1)
  int64_t num_users, num_items;
  const float* w;        // (ndense) weight of the dense columns
  const float* b;        // (1)
};

struct WideGrads {
  float* user1;
  float* item1;
  float* w;
  float* b;
};

__device__ __forceinline__ int64_t clamp_row(float v, int64_t n, int32_t* err) {
  int64_t r = (int64_t)v;
  if (r < 0 || r >= n) {
    if (err) *err = 1;
    r = 0;
  }
  return r;
}

// -------------------------------------------------------------- DeepFM wide + FM
// out[b] = ((user1[u] + item1[i]) + (x[b,dense] . w + wb)) + 0.5 * sum_e[(sum_f v)^2 - sum_f v^2]
// (model/deepfm.py:63, 71-77)
template <int VEC>
__global__ void __launch_bounds__(kBlock)
fm_wide_fwd_kernel(const Geometry g, const float* __restrict__ emb, int64_t lde, int64_t batch, const WideArgs a,
                   float* __restrict__ out, int64_t ldo, int32_t* err) {
  extern __shared__ float lds[];
  float* s_part = lds + g.tb * g.row;  // [tb][chunks] partial FM sums
  const int chunks = g.e / VEC;
  for (int64_t b0 = (int64_t)blockIdx.x * g.tb; b0 < batch; b0 += (int64_t)gridDim.x * g.tb) {
    const int count = (int)(batch - b0 < g.tb ? batch - b0 : g.tb);
    stage_vectors<VEC>(lds, g, emb, lde, b0, count);
    __syncthreads();
    for (int w = threadIdx.x; w < count * chunks; w += blockDim.x) {
      const int s = w / chunks, c = (w - s * chunks) * VEC;
      const float* base = lds + s * g.row + c;
      float part = 0.0f;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float sum = 0.0f, sq = 0.0f;
        for (int f = 0; f < g.nvec; ++f) {
          const float t = base[f * g.estride + v];
          sum += t;
          sq = fmaf(t, t, sq);
        }
        part += sum * sum - sq;
      }
      s_part[s * chunks + (c / VEC)] = part;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < count; s += blockDim.x) {
      float cross = 0.0f;
      for (int c = 0; c < chunks; ++c) cross += s_part[s * chunks + c];
      cross *= 0.5f;
      const float* xr = a.x + (b0 + s) * a.ldx;
      const int64_t u = clamp_row(xr[a.user_col], a.num_users, err);
      const int64_t i = clamp_row(xr[a.item_col], a.num_items, err);
      float lin = 0.0f;
      for (int c = 0; c < a.ndense; ++c) lin = fmaf(xr[a.dense_col0 + c], a.w[c], lin);
      lin += a.b[0];
      out[(b0 + s) * ldo] = ((a.user1[u] + a.item1[i]) + lin) + cross;
    }
    __syncthreads();
  }
}

// backward of the above for g = gout[b]:
//   user1[u] += g, item1[i] += g, w[c] += sum_b g x[b,c], b += sum_b g,
//   gemb[b,f,e] (+)= g * (S_e - v_fe)
template <int VEC>
__global__ void __launch_bounds__(kBlock)
fm_wide_bwd_kernel(const Geometry g, const float* __restrict__ emb, int64_t lde, int64_t batch, const WideArgs a,
                   const float* __restrict__ gout, int64_t ldgo, const WideGrads wg, float* __restrict__ gemb,
                   int64_t ldg, int accumulate, float* __restrict__ ws) {
  extern __shared__ float lds[];
  __shared__ float s_w[kMaxDense + 1];  // block-level sums for w and b
  float* s_g = lds + g.tb * g.row;      // [tb]
  float* s_x = s_g + g.tb;              // [tb][ndense]: the dense feature columns of the tile
  const int chunks = g.e / VEC;
  for (int i = threadIdx.x; i <= a.ndense; i += blockDim.x) s_w[i] = 0.0f;
  for (int64_t b0 = (int64_t)blockIdx.x * g.tb; b0 < batch; b0 += (int64_t)gridDim.x * g.tb) {
    const int count = (int)(batch - b0 < g.tb ? batch - b0 : g.tb);
    stage_vectors<VEC>(lds, g, emb, lde, b0, count);
    for (int s = threadIdx.x; s < count; s += blockDim.x) s_g[s] = gout[(b0 + s) * ldgo];
    // all 256 threads fetch the tile's dense columns (coalesced): the weight-gradient loop below
    // used to read them from HBM with 44 threads, one dependent load per sample
    for (int i = threadIdx.x; i < count * a.ndense; i += blockDim.x) {
      const int s = i / a.ndense, c = i - s * a.ndense;
      s_x[i] = a.x[(b0 + s) * a.ldx + a.dense_col0 + c];
    }
    __syncthreads();
    // per-sample scalars: id tables
    for (int s = threadIdx.x; s < count; s += blockDim.x) {
      const float gv = s_g[s];
      const float* xr = a.x + (b0 + s) * a.ldx;
      int64_t u = (int64_t)xr[a.user_col], i = (int64_t)xr[a.item_col];
      if (u < 0 || u >= a.num_users) u = 0;
      if (i < 0 || i >= a.num_items) i = 0;
      if (wg.user1) unsafeAtomicAdd(wg.user1 + u, gv);
      if (wg.item1) unsafeAtomicAdd(wg.item1 + i, gv);
    }
    // dense weights: thread c sums g*x over the tile (sequential over samples: fixed order)
    if (wg.w || wg.b) {
      for (int c = threadIdx.x; c <= a.ndense; c += blockDim.x) {
        float acc = 0.0f;
        for (int s = 0; s < count; ++s) {
          const float xv = c < a.ndense ? s_x[s * a.ndense + c] : 1.0f;
          acc = fmaf(s_g[s], xv, acc);
        }
        s_w[c] += acc;
      }
    }
    // FM second-order gradient
    if (gemb) {
      const int per_sample = g.nvec * chunks;
      for (int w = threadIdx.x; w < count * per_sample; w += blockDim.x) {
        const int s = w / per_sample, rem = w - s * per_sample;
        const int f = rem / chunks, c = (rem - f * chunks) * VEC;
        const float* base = lds + s * g.row + c;
        float* dst = gemb + (b0 + s) * ldg + f * g.e + c;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float sum = 0.0f;
          for (int ff = 0; ff < g.nvec; ++ff) sum += base[ff * g.estride + v];
          const float val = s_g[s] * (sum - base[f * g.estride + v]);
          dst[v] = accumulate ? dst[v] + val : val;
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  for (int c = threadIdx.x; c <= a.ndense; c += blockDim.x) {
    const float v = s_w[c];
    if (ws) {
      ws[(int64_t)blockIdx.x * (a.ndense + 1) + c] = v;  // summed by reduce.hip
    } else if (c < a.ndense) {
      if (wg.w && v != 0.0f) unsafeAtomicAdd(wg.w + c, v);
    } else if (wg.b && v != 0.0f) {
      unsafeAtomicAdd(wg.b, v);
    }
  }
}

// ---------------------------------------------------------------- FFM head
struct PairList {
  int n;
  unsigned char a[kMaxPairs];
  unsigned char b[kMaxPairs];
  // backward view: vector f pairs with partner[start[f] .. start[f+1])
  unsigned char start[65];
  unsigned char partner[2 * kMaxPairs];
};

// prob[b] = sigmoid(user1[u] + item1[i] + sum_c (x[b,c] + cross) w[c] + wb),
// cross = sum_p <v_a(p), v_b(p)> added left to right (model/ffm.py:62-86; the
// cross scalar is added to every dense input before the linear layer, as there)
template <int VEC>
__global__ void __launch_bounds__(kBlock)
ffm_head_fwd_kernel(const Geometry g, const float* __restrict__ emb, int64_t lde, int64_t batch, const PairList pl,
                    const WideArgs a, float* __restrict__ prob, int64_t ldo, int32_t* err) {
  extern __shared__ float lds[];
  float* s_dot = lds + g.tb * g.row;  // [tb][npairs]
  for (int64_t b0 = (int64_t)blockIdx.x * g.tb; b0 < batch; b0 += (int64_t)gridDim.x * g.tb) {
    const int count = (int)(batch - b0 < g.tb ? batch - b0 : g.tb);
    stage_vectors<VEC>(lds, g, emb, lde, b0, count);
    __syncthreads();
    for (int w = threadIdx.x; w < count * pl.n; w += blockDim.x) {
      const int s = w / pl.n, p = w - s * pl.n;
      const float* base = lds + s * g.row;
      s_dot[s * pl.n + p] = dot_lds<VEC>(base + pl.a[p] * g.estride, base + pl.b[p] * g.estride, g.e);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < count; s += blockDim.x) {
      float cross = s_dot[s * pl.n];
      for (int p = 1; p < pl.n; ++p) cross += s_dot[s * pl.n + p];
      const float* xr = a.x + (b0 + s) * a.ldx;
      const int64_t u = clamp_row(xr[a.user_col], a.num_users, err);
      const int64_t i = clamp_row(xr[a.item_col], a.num_items, err);
      float lin = 0.0f;
      for (int c = 0; c < a.ndense; ++c) lin = fmaf(xr[a.dense_col0 + c] + cross, a.w[c], lin);
      lin += a.b[0];
      prob[(b0 + s) * ldo] = ctr_sigmoid((a.user1[u] + a.item1[i]) + lin);
    }
    __syncthreads();
  }
}

template <int VEC>
__global__ void __launch_bounds__(kBlock)
ffm_head_bwd_kernel(const Geometry g, const float* __restrict__ emb, int64_t lde, int64_t batch, const PairList pl,
                    const WideArgs a, const float* __restrict__ prob, int64_t ldp, const float* __restrict__ gprob,
                    int64_t ldgp, const WideGrads wg, float* __restrict__ gemb, int64_t ldg,
                    float* __restrict__ ws) {
  extern __shared__ float lds[];
  __shared__ float s_w[kMaxDense + 1];
  __shared__ float s_wsum;
  float* s_dot = lds + g.tb * g.row;      // [tb][npairs] then reused
  float* s_dz = s_dot + g.tb * pl.n;      // [tb] dlogit
  float* s_cross = s_dz + g.tb;           // [tb]
  float* s_x = s_cross + g.tb;            // [tb][ndense]: the dense feature columns of the tile
  const int chunks = g.e / VEC;
  for (int i = threadIdx.x; i <= a.ndense; i += blockDim.x) s_w[i] = 0.0f;
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int c = 0; c < a.ndense; ++c) t += a.w[c];
    s_wsum = t;
  }
  for (int64_t b0 = (int64_t)blockIdx.x * g.tb; b0 < batch; b0 += (int64_t)gridDim.x * g.tb) {
    const int count = (int)(batch - b0 < g.tb ? batch - b0 : g.tb);
    stage_vectors<VEC>(lds, g, emb, lde, b0, count);
    for (int i = threadIdx.x; i < count * a.ndense; i += blockDim.x) {  // see fm_wide_bwd_kernel
      const int s = i / a.ndense, c = i - s * a.ndense;
      s_x[i] = a.x[(b0 + s) * a.ldx + a.dense_col0 + c];
    }
    __syncthreads();
    for (int w = threadIdx.x; w < count * pl.n; w += blockDim.x) {
      const int s = w / pl.n, p = w - s * pl.n;
      const float* base = lds + s * g.row;
      s_dot[s * pl.n + p] = dot_lds<VEC>(base + pl.a[p] * g.estride, base + pl.b[p] * g.estride, g.e);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < count; s += blockDim.x) {
      float cross = s_dot[s * pl.n];
      for (int p = 1; p < pl.n; ++p) cross += s_dot[s * pl.n + p];
      const float pv = prob[(b0 + s) * ldp];
      const float dz = gprob[(b0 + s) * ldgp] * pv * (1.0f - pv);
      s_dz[s] = dz;
      s_cross[s] = cross;
      const float* xr = a.x + (b0 + s) * a.ldx;
      int64_t u = (int64_t)xr[a.user_col], i = (int64_t)xr[a.item_col];
      if (u < 0 || u >= a.num_users) u = 0;
      if (i < 0 || i >= a.num_items) i = 0;
      if (wg.user1) unsafeAtomicAdd(wg.user1 + u, dz);
      if (wg.item1) unsafeAtomicAdd(wg.item1 + i, dz);
    }
    __syncthreads();
    if (wg.w || wg.b) {
      for (int c = threadIdx.x; c <= a.ndense; c += blockDim.x) {
        float acc = 0.0f;
        for (int s = 0; s < count; ++s) {
          const float xv = c < a.ndense ? s_x[s * a.ndense + c] + s_cross[s] : 1.0f;
          acc = fmaf(s_dz[s], xv, acc);
        }
        s_w[c] += acc;
      }
    }
    if (gemb) {
      const int per_sample = g.nvec * chunks;
      for (int w = threadIdx.x; w < count * per_sample; w += blockDim.x) {
        const int s = w / per_sample, rem = w - s * per_sample;
        const int f = rem / chunks, c = (rem - f * chunks) * VEC;
        const float* base = lds + s * g.row + c;
        const float dcross = s_dz[s] * s_wsum;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
        for (int q = pl.start[f]; q < pl.start[f + 1]; ++q) {
          const int other = pl.partner[q];
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] += base[other * g.estride + v];
        }
        float* dst = gemb + (b0 + s) * ldg + f * g.e + c;
#pragma unroll
        for (int v = 0; v < VEC; ++v) dst[v] = dcross * acc[v];
      }
    }
    __syncthreads();
  }
  __syncthreads();
  for (int c = threadIdx.x; c <= a.ndense; c += blockDim.x) {
    const float v = s_w[c];
    if (ws) {
      ws[(int64_t)blockIdx.x * (a.ndense + 1) + c] = v;  // summed by reduce.hip
    } else if (c < a.ndense) {
      if (wg.w && v != 0.0f) unsafeAtomicAdd(wg.w + c, v);
    } else if (wg.b && v != 0.0f) {
      unsafeAtomicAdd(wg.b, v);
    }
  }
}

// ------------------------------------------------------------ activation bwd
__global__ void __launch_bounds__(kBlock)
act_bwd_kernel(const float* __restrict__ y, int64_t ldy, const float* __restrict__ gy, int64_t ldgy,
               float* __restrict__ out, int64_t ldo, int64_t m, int n, int act, int accumulate) {
  const int64_t total = m * n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / n;
    const int c = (int)(t - r * n);
    const float v = gy[r * ldgy + c] * ctr_act_grad(y[r * ldy + c], act);
    float* dst = out + r * ldo + c;
    *dst = accumulate ? *dst + v : v;
  }
}

int check_wide(const WideArgs& a) {
  CTR_REQUIRE(a.x && a.ldx > 0 && a.user1 && a.item1 && a.w && a.b, CTR_EINVAL);
  CTR_REQUIRE(a.num_users > 0 && a.num_items > 0, CTR_EINVAL);
  CTR_REQUIRE(a.ndense > 0 && a.ndense <= kMaxDense, CTR_ELIMIT);
  CTR_REQUIRE(a.user_col >= 0 && a.item_col >= 0 && a.dense_col0 >= 0 && a.dense_col0 + a.ndense <= a.ldx &&
                  a.user_col < a.ldx && a.item_col < a.ldx,
              CTR_EINVAL);
  return CTR_OK;
}

// the dense-column weight/bias gradients are one (ndense+1)-float partial per workgroup:
// through the workspace when there is one (stores + reduce.hip), else short atomic chains
inline float* small_grad_workspace(float* workspace, int64_t floats, int ndense, bool wanted, int* grid) {
  if (*grid > 1024) *grid = 1024;
  if (!wanted) return nullptr;
  if (workspace && floats >= (int64_t)(*grid) * (ndense + 1)) return workspace;
  if (*grid > 128) *grid = 128;
  return nullptr;
}

inline int reduce_small_grads(const float* ws, int parts, int ndense, float* gw, float* gb, hipStream_t st) {
  CtrSegments segs;
  segs.n = 0;
  if (gw) segs.s[segs.n++] = CtrSegment{0, ndense, gw};
  if (gb) segs.s[segs.n++] = CtrSegment{ndense, 1, gb};
  return ctr_reduce_segments(ws, parts, ndense + 1, segs, st);
}

inline int tile_grid(int64_t batch, int tb) {
  int64_t tiles = ctr_ceil_div(batch, tb);
  return (int)(tiles < 2048 ? tiles : 2048);
}

}  // namespace

extern "C" int ctr_allpairs_fwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim, float* out,
                                int64_t ldo, void* stream) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(emb && out && nvec >= 2 && nvec <= 32 && dim > 0, CTR_EINVAL);
  CTR_REQUIRE(lde >= (int64_t)nvec * dim && ldo >= nvec * (nvec - 1) / 2, CTR_EINVAL);
  const Geometry g = make_geometry(nvec, dim, ctr_aligned16(emb) && lde % 4 == 0);
  CTR_REQUIRE(g.tb >= 1, CTR_ELIMIT);
  const size_t dyn = (size_t)g.tb * g.row * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (g.vec == 4)
    hipLaunchKernelGGL(allpairs_fwd_kernel<4>, dim3(tile_grid(batch, g.tb)), dim3(kBlock), dyn, st, g, emb, lde, batch,
                       out, ldo);
  else
    hipLaunchKernelGGL(allpairs_fwd_kernel<1>, dim3(tile_grid(batch, g.tb)), dim3(kBlock), dyn, st, g, emb, lde, batch,
                       out, ldo);
  return ctr_launch_status();
}

extern "C" int ctr_allpairs_bwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim, const float* gp,
                                int64_t ldgp, float* gemb, int64_t ldg, int accumulate, void* stream) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(emb && gp && gemb && nvec >= 2 && nvec <= 32 && dim > 0, CTR_EINVAL);
  const int npairs = nvec * (nvec - 1) / 2;
  CTR_REQUIRE(lde >= (int64_t)nvec * dim && ldg >= (int64_t)nvec * dim && ldgp >= npairs, CTR_EINVAL);
  const Geometry g = make_geometry(nvec, dim, ctr_aligned16(emb) && lde % 4 == 0, npairs);
  CTR_REQUIRE(g.tb >= 1, CTR_ELIMIT);
  const size_t dyn = (size_t)g.tb * (g.row + npairs) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (g.vec == 4)
    hipLaunchKernelGGL(allpairs_bwd_kernel<4>, dim3(tile_grid(batch, g.tb)), dim3(kBlock), dyn, st, g, emb, lde, batch,
                       gp, ldgp, gemb, ldg, accumulate);
  else
    hipLaunchKernelGGL(allpairs_bwd_kernel<1>, dim3(tile_grid(batch, g.tb)), dim3(kBlock), dyn, st, g, emb, lde, batch,
                       gp, ldgp, gemb, ldg, accumulate);
  return ctr_launch_status();
}

static WideArgs wide_args(const float* x, int64_t ldx, int user_col, int item_col, int dense_col0, int ndense,
                          const float* user1, int64_t num_users, const float* item1, int64_t num_items,
                          const float* w, const float* b) {
  WideArgs a;
  a.x = x; a.ldx = ldx; a.user_col = user_col; a.item_col = item_col; a.dense_col0 = dense_col0; a.ndense = ndense;
  a.user1 = user1; a.item1 = item1; a.num_users = num_users; a.num_items = num_items; a.w = w; a.b = b;
  return a;
}

extern "C" int ctr_fm_wide_fwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim, const float* x,
                               int64_t ldx, int user_col, int item_col, int dense_col0, int ndense,
                               const float* user1, int64_t num_users, const float* item1, int64_t num_items,
                               const float* wide_w, const float* wide_b, float* out, int64_t ldo, int32_t* err_flag,
                               void* stream) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(emb && out && nvec >= 1 && nvec <= 64 && dim > 0 && ldo >= 1 && lde >= (int64_t)nvec * dim, CTR_EINVAL);
  const WideArgs a = wide_args(x, ldx, user_col, item_col, dense_col0, ndense, user1, num_users, item1, num_items,
                               wide_w, wide_b);
  int rc = check_wide(a);
  if (rc != CTR_OK) return rc;
  const bool al = ctr_aligned16(emb) && lde % 4 == 0;
  const int chunks = (al && dim % 4 == 0) ? dim / 4 : dim;
  const Geometry g = make_geometry(nvec, dim, al, chunks);
  CTR_REQUIRE(g.tb >= 1, CTR_ELIMIT);
  const size_t dyn = (size_t)g.tb * (g.row + chunks) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (g.vec == 4)
    hipLaunchKernelGGL(fm_wide_fwd_kernel<4>, dim3(tile_grid(batch, g.tb)), dim3(kBlock), dyn, st, g, emb, lde, batch, a,
                       out, ldo, err_flag);
  else
    hipLaunchKernelGGL(fm_wide_fwd_kernel<1>, dim3(tile_grid(batch, g.tb)), dim3(kBlock), dyn, st, g, emb, lde, batch, a,
                       out, ldo, err_flag);
  return ctr_launch_status();
}

extern "C" int ctr_fm_wide_bwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim, const float* x,
                               int64_t ldx, int user_col, int item_col, int dense_col0, int ndense,
                               const float* user1, int64_t num_users, const float* item1, int64_t num_items,
                               const float* wide_w, const float* wide_b, const float* gout, int64_t ldgo,
                               float* guser1, float* gitem1, float* gwide_w, float* gwide_b, float* gemb, int64_t ldg,
                               int accumulate, float* workspace, int64_t workspace_floats, void* stream) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(emb && gout && nvec >= 1 && nvec <= 64 && dim > 0 && ldgo >= 1 && lde >= (int64_t)nvec * dim, CTR_EINVAL);
  CTR_REQUIRE(!gemb || ldg >= (int64_t)nvec * dim, CTR_EINVAL);
  const WideArgs a = wide_args(x, ldx, user_col, item_col, dense_col0, ndense, user1, num_users, item1, num_items,
                               wide_w, wide_b);
  int rc = check_wide(a);
  if (rc != CTR_OK) return rc;
  const Geometry g = make_geometry(nvec, dim, ctr_aligned16(emb) && lde % 4 == 0, 1 + ndense);
  CTR_REQUIRE(g.tb >= 1, CTR_ELIMIT);
  const size_t dyn = (size_t)g.tb * (g.row + 1 + ndense) * sizeof(float);
  const WideGrads wg{guser1, gitem1, gwide_w, gwide_b};
  int grid = tile_grid(batch, g.tb);
  float* ws = small_grad_workspace(workspace, workspace_floats, ndense, gwide_w || gwide_b, &grid);
  hipStream_t st = (hipStream_t)stream;
  if (g.vec == 4)
    hipLaunchKernelGGL(fm_wide_bwd_kernel<4>, dim3(grid), dim3(kBlock), dyn, st, g, emb, lde, batch, a, gout, ldgo, wg,
                       gemb, ldg, accumulate, ws);
  else
    hipLaunchKernelGGL(fm_wide_bwd_kernel<1>, dim3(grid), dim3(kBlock), dyn, st, g, emb, lde, batch, a, gout, ldgo, wg,
                       gemb, ldg, accumulate, ws);
  rc = ctr_launch_status();
  if (rc != CTR_OK || !ws) return rc;
  return reduce_small_grads(ws, grid, ndense, gwide_w, gwide_b, st);
}

static int make_pairs(const int32_t* pairs, int npairs, int nvec, PairList* pl) {
  CTR_REQUIRE(pairs && npairs >= 1 && npairs <= kMaxPairs, CTR_ELIMIT);
  pl->n = npairs;
  for (int p = 0; p < npairs; ++p) {
    const int a = pairs[2 * p], b = pairs[2 * p + 1];
    CTR_REQUIRE(a >= 0 && a < nvec && b >= 0 && b < nvec && a != b, CTR_EINVAL);
    pl->a[p] = (unsigned char)a;
    pl->b[p] = (unsigned char)b;
  }
  int q = 0;
  for (int f = 0; f < nvec && f < 64; ++f) {
    pl->start[f] = (unsigned char)q;
    for (int p = 0; p < npairs; ++p) {
      if (pairs[2 * p] == f) pl->partner[q++] = (unsigned char)pairs[2 * p + 1];
      else if (pairs[2 * p + 1] == f) pl->partner[q++] = (unsigned char)pairs[2 * p];
    }
  }
  for (int f = nvec; f <= 64; ++f) pl->start[f] = (unsigned char)q;
  pl->start[nvec < 64 ? nvec : 64] = (unsigned char)q;
  return CTR_OK;
}

extern "C" int ctr_ffm_head_fwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim,
                                const int32_t* pairs /*host, 2*npairs*/, int npairs, const float* x, int64_t ldx,
                                int user_col, int item_col, int dense_col0, int ndense, const float* user1,
                                int64_t num_users, const float* item1, int64_t num_items, const float* lin_w,
                                const float* lin_b, float* prob, int64_t ldo, int32_t* err_flag, void* stream) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(emb && prob && nvec >= 2 && nvec <= 64 && dim > 0 && ldo >= 1 && lde >= (int64_t)nvec * dim, CTR_EINVAL);
  PairList pl;
  int rc = make_pairs(pairs, npairs, nvec, &pl);
  if (rc != CTR_OK) return rc;
  const WideArgs a = wide_args(x, ldx, user_col, item_col, dense_col0, ndense, user1, num_users, item1, num_items,
                               lin_w, lin_b);
  rc = check_wide(a);
  if (rc != CTR_OK) return rc;
  const Geometry g = make_geometry(nvec, dim, ctr_aligned16(emb) && lde % 4 == 0, npairs);
  CTR_REQUIRE(g.tb >= 1, CTR_ELIMIT);
  const size_t dyn = (size_t)g.tb * (g.row + npairs) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (g.vec == 4)
    hipLaunchKernelGGL(ffm_head_fwd_kernel<4>, dim3(tile_grid(batch, g.tb)), dim3(kBlock), dyn, st, g, emb, lde, batch,
                       pl, a, prob, ldo, err_flag);
  else
    hipLaunchKernelGGL(ffm_head_fwd_kernel<1>, dim3(tile_grid(batch, g.tb)), dim3(kBlock), dyn, st, g, emb, lde, batch,
                       pl, a, prob, ldo, err_flag);
  return ctr_launch_status();
}

extern "C" int ctr_ffm_head_bwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim, const int32_t* pairs,
                                int npairs, const float* x, int64_t ldx, int user_col, int item_col, int dense_col0,
                                int ndense, const float* user1, int64_t num_users, const float* item1,
                                int64_t num_items, const float* lin_w, const float* lin_b, const float* prob,
                                int64_t ldp, const float* gprob, int64_t ldgp, float* guser1, float* gitem1,
                                float* glin_w, float* glin_b, float* gemb, int64_t ldg, float* workspace,
                                int64_t workspace_floats, void* stream) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(emb && prob && gprob && nvec >= 2 && nvec <= 64 && dim > 0 && lde >= (int64_t)nvec * dim, CTR_EINVAL);
  CTR_REQUIRE(ldp >= 1 && ldgp >= 1 && (!gemb || ldg >= (int64_t)nvec * dim), CTR_EINVAL);
  PairList pl;
  int rc = make_pairs(pairs, npairs, nvec, &pl);
  if (rc != CTR_OK) return rc;
  const WideArgs a = wide_args(x, ldx, user_col, item_col, dense_col0, ndense, user1, num_users, item1, num_items,
                               lin_w, lin_b);
  rc = check_wide(a);
  if (rc != CTR_OK) return rc;
  const Geometry g = make_geometry(nvec, dim, ctr_aligned16(emb) && lde % 4 == 0, npairs + 2 + ndense);
  CTR_REQUIRE(g.tb >= 1, CTR_ELIMIT);
  const size_t dyn = (size_t)g.tb * (g.row + npairs + 2 + ndense) * sizeof(float);
  const WideGrads wg{guser1, gitem1, glin_w, glin_b};
  int grid = tile_grid(batch, g.tb);
  float* ws = small_grad_workspace(workspace, workspace_floats, ndense, glin_w || glin_b, &grid);
  hipStream_t st = (hipStream_t)stream;
  if (g.vec == 4)
    hipLaunchKernelGGL(ffm_head_bwd_kernel<4>, dim3(grid), dim3(kBlock), dyn, st, g, emb, lde, batch, pl, a, prob, ldp,
                       gprob, ldgp, wg, gemb, ldg, ws);
  else
    hipLaunchKernelGGL(ffm_head_bwd_kernel<1>, dim3(grid), dim3(kBlock), dyn, st, g, emb, lde, batch, pl, a, prob, ldp,
                       gprob, ldgp, wg, gemb, ldg, ws);
  rc = ctr_launch_status();
  if (rc != CTR_OK || !ws) return rc;
  return reduce_small_grads(ws, grid, ndense, glin_w, glin_b, st);
}

extern "C" int ctr_act_bwd(const float* y, int64_t ldy, const float* gy, int64_t ldgy, float* out, int64_t ldo,
                           int64_t m, int n, int act, int accumulate, void* stream) {
  CTR_REQUIRE(m >= 0 && n > 0, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  CTR_REQUIRE(y && gy && out && ldy >= n && ldgy >= n && ldo >= n, CTR_EINVAL);
  CTR_REQUIRE(act >= CTR_ACT_NONE && act <= CTR_ACT_SIGMOID, CTR_EINVAL);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ctr_stream_grid(m * n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, y, ldy,
                     gy, ldgy, out, ldo, m, n, act, accumulate);
  return ctr_launch_status();
}


// ---------------------------------------------------------------------------
// NFM bi-interaction pooling (model/nfm.py:56-61): out[b, e] = sum_{i<j} v_i[e] * v_j[e] over the
// nvec vectors of a sample, summed in the reference's pair order (the (sum v)^2 - sum v^2 identity
// would cancel digits the parity bar wants).  Backward: gemb[b, i, e] (= or +=) g[b, e] * (S[e] -
// v_i[e]).  Streaming: a lane owns one (sample, element) column.
namespace {

constexpr int kBiBlock = 256;

// VEC = 4: a thread owns 4 consecutive elements of a sample (16-byte loads / stores) when dim % 4 == 0 and every
// row is 16-byte aligned; the per-element arithmetic and its order are the same as with VEC = 1.
template <int VEC>
struct BiVec {
  float v[VEC];
  __device__ __forceinline__ void load(const float* p) {
    if (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[VEC - 1] = t.w;
    } else {
      v[0] = *p;
    }
  }
  __device__ __forceinline__ void store(float* p) const {
    if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[VEC - 1]);
    else *p = v[0];
  }
};

template <int VEC>
__global__ void __launch_bounds__(kBiBlock)
biinteract_fwd_kernel(const float* __restrict__ emb, int64_t lde, int64_t batch, int nvec, int dim,
                      float* __restrict__ out, int64_t ldo) {
  const int dv = dim / VEC;
  const int64_t total = batch * dv;
  for (int64_t g = (int64_t)blockIdx.x * kBiBlock + threadIdx.x; g < total; g += (int64_t)gridDim.x * kBiBlock) {
    const int64_t b = g / dv;
    const int e = (int)(g - b * dv) * VEC;
    const float* v = emb + b * lde + e;
    BiVec<VEC> acc;
#pragma unroll
    for (int u = 0; u < VEC; ++u) acc.v[u] = 0.0f;
    for (int i = 0; i < nvec; ++i) {
      BiVec<VEC> vi;
      vi.load(v + (int64_t)i * dim);
      for (int j = i + 1; j < nvec; ++j) {
        BiVec<VEC> vj;
        vj.load(v + (int64_t)j * dim);
#pragma unroll
        for (int u = 0; u < VEC; ++u) acc.v[u] += vi.v[u] * vj.v[u];
      }
    }
    acc.store(out + b * ldo + e);
  }
}

template <int VEC>
__global__ void __launch_bounds__(kBiBlock)
biinteract_bwd_kernel(const float* __restrict__ emb, int64_t lde, int64_t batch, int nvec, int dim,
                      const float* __restrict__ gout, int64_t ldgo, float* __restrict__ gemb, int64_t ldg,
                      int accumulate) {
  const int dv = dim / VEC;
  const int64_t total = batch * dv;
  for (int64_t g = (int64_t)blockIdx.x * kBiBlock + threadIdx.x; g < total; g += (int64_t)gridDim.x * kBiBlock) {
    const int64_t b = g / dv;
    const int e = (int)(g - b * dv) * VEC;
    const float* v = emb + b * lde + e;
    BiVec<VEC> sum, go;
#pragma unroll
    for (int u = 0; u < VEC; ++u) sum.v[u] = 0.0f;
    for (int i = 0; i < nvec; ++i) {
      BiVec<VEC> vi;
      vi.load(v + (int64_t)i * dim);
#pragma unroll
      for (int u = 0; u < VEC; ++u) sum.v[u] += vi.v[u];
    }
    go.load(gout + b * ldgo + e);
    float* o = gemb + b * ldg + e;
    for (int i = 0; i < nvec; ++i) {
      BiVec<VEC> vi, val;
      vi.load(v + (int64_t)i * dim);
#pragma unroll
      for (int u = 0; u < VEC; ++u) val.v[u] = go.v[u] * (sum.v[u] - vi.v[u]);
      if (accumulate) {
        BiVec<VEC> old;
        old.load(o + (int64_t)i * dim);
#pragma unroll
        for (int u = 0; u < VEC; ++u) val.v[u] = old.v[u] + val.v[u];
      }
      val.store(o + (int64_t)i * dim);
    }
  }
}

}  // namespace

extern "C" int ctr_biinteract_fwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim, float* out,
                                  int64_t ldo, void* stream) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(emb && out && nvec >= 1 && nvec <= 64 && dim > 0 && lde >= (int64_t)nvec * dim && ldo >= dim, CTR_EINVAL);
  if (dim % 4 == 0 && lde % 4 == 0 && ldo % 4 == 0 && ctr_aligned16(emb) && ctr_aligned16(out))
    hipLaunchKernelGGL(biinteract_fwd_kernel<4>, dim3(ctr_stream_grid(batch * (dim / 4), kBiBlock)), dim3(kBiBlock), 0,
                       (hipStream_t)stream, emb, lde, batch, nvec, dim, out, ldo);
  else
    hipLaunchKernelGGL(biinteract_fwd_kernel<1>, dim3(ctr_stream_grid(batch * dim, kBiBlock)), dim3(kBiBlock), 0,
                       (hipStream_t)stream, emb, lde, batch, nvec, dim, out, ldo);
  return ctr_launch_status();
}

extern "C" int ctr_biinteract_bwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim, const float* gout,
                                  int64_t ldgo, float* gemb, int64_t ldg, int accumulate, void* stream) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(emb && gout && gemb && nvec >= 1 && nvec <= 64 && dim > 0, CTR_EINVAL);
  CTR_REQUIRE(lde >= (int64_t)nvec * dim && ldg >= (int64_t)nvec * dim && ldgo >= dim, CTR_EINVAL);
  if (dim % 4 == 0 && lde % 4 == 0 && ldgo % 4 == 0 && ldg % 4 == 0 && ctr_aligned16(emb) && ctr_aligned16(gout) &&
      ctr_aligned16(gemb))
    hipLaunchKernelGGL(biinteract_bwd_kernel<4>, dim3(ctr_stream_grid(batch * (dim / 4), kBiBlock)), dim3(kBiBlock), 0,
                       (hipStream_t)stream, emb, lde, batch, nvec, dim, gout, ldgo, gemb, ldg, accumulate);
  else
    hipLaunchKernelGGL(biinteract_bwd_kernel<1>, dim3(ctr_stream_grid(batch * dim, kBiBlock)), dim3(kBiBlock), 0,
                       (hipStream_t)stream, emb, lde, batch, nvec, dim, gout, ldgo, gemb, ldg, accumulate);
  return ctr_launch_status();
}


// ---------------------------------------------------------------------------
// AFM pair products (model/afm.py:56-60): out[(b*np + idx(i,j)), e] = v_i[e] * v_j[e], i < j in
// lexicographic order, np = nvec*(nvec-1)/2 rows per sample -- the operand of the attention
// net.  Backward: the pair gradient is gp (through the attention net) plus, when given,
// attn[b, p] * gpool[b, :] (through the attention-weighted sum, afm.py:65); then
// gemb[b, i, e] (= or +=) sum_{j != i} gpair[b, idx(i,j), e] * v_j[e].
namespace {

// NV = number of vectors as a compile-time constant (2..8 instantiated, 16 = generic with guards): the
// vectors of a sample element live in registers and the pair loops unroll.  With a run-time nvec the
// backward's acc[i] / acc[j] were dynamically indexed, i.e. in scratch memory, and the forward re-read
// v_j for every pair (AFM, 5 vectors: 230 / 294 us for 335 MB of pair products).
template <int NV>
__global__ void __launch_bounds__(kBiBlock)
pairprod_fwd_kernel(const float* __restrict__ emb, int64_t lde, int64_t batch, int nvec, int dim,
                    float* __restrict__ out, int64_t ldo) {
  const int np = nvec * (nvec - 1) / 2;
  const int64_t total = batch * dim;
  for (int64_t g = (int64_t)blockIdx.x * kBiBlock + threadIdx.x; g < total; g += (int64_t)gridDim.x * kBiBlock) {
    const int64_t b = g / dim;
    const int e = (int)(g - b * dim);
    const float* v = emb + b * lde + e;
    float* o = out + b * np * ldo + e;
    float vv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) vv[i] = i < nvec ? v[(int64_t)i * dim] : 0.0f;
    int p = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = i + 1; j < NV; ++j)
        if (j < nvec) {
          o[(int64_t)p * ldo] = vv[i] * vv[j];
          ++p;
        }
  }
}

template <int NV, int VEC>
__global__ void __launch_bounds__(kBiBlock)
pairprod_bwd_kernel(const float* __restrict__ emb, int64_t lde, int64_t batch, int nvec, int dim,
                    const float* __restrict__ gp, int64_t ldgp, const float* __restrict__ attn,
                    const float* __restrict__ gpool, int64_t ldgo, float* __restrict__ gemb, int64_t ldg,
                    int accumulate) {
  // VEC = 4: a thread owns 4 consecutive elements (16-byte loads / stores; dim % 4 == 0, aligned rows)
  const int np = nvec * (nvec - 1) / 2;
  const int dv = dim / VEC;
  const int64_t total = batch * dv;
  const bool pooled = attn && gpool;
  for (int64_t g = (int64_t)blockIdx.x * kBiBlock + threadIdx.x; g < total; g += (int64_t)gridDim.x * kBiBlock) {
    const int64_t b = g / dv;
    const int e = (int)(g - b * dv) * VEC;
    const float* v = emb + b * lde + e;
    const float* q = gp + b * np * ldgp + e;
    float go[VEC], vv[NV][VEC], acc[NV][VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) go[u] = 0.0f;
    if (pooled) {
      if (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(gpool + b * ldgo + e);
        go[0] = t.x; go[1] = t.y; go[2] = t.z; go[VEC - 1] = t.w;
      } else {
        go[0] = gpool[b * ldgo + e];
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int u = 0; u < VEC; ++u) vv[i][u] = acc[i][u] = 0.0f;
      if (i < nvec) {
        if (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(v + (int64_t)i * dim);
          vv[i][0] = t.x; vv[i][1] = t.y; vv[i][2] = t.z; vv[i][VEC - 1] = t.w;
        } else {
          vv[i][0] = v[(int64_t)i * dim];
        }
      }
    }
    int p = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = i + 1; j < NV; ++j)
        if (j < nvec) {
          float gpair[VEC];
          if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(q + (int64_t)p * ldgp);
            gpair[0] = t.x; gpair[1] = t.y; gpair[2] = t.z; gpair[VEC - 1] = t.w;
          } else {
            gpair[0] = q[(int64_t)p * ldgp];
          }
          const float at = pooled ? attn[b * np + p] : 0.0f;
#pragma unroll
          for (int u = 0; u < VEC; ++u) {
            if (pooled) gpair[u] = fmaf(at, go[u], gpair[u]);
            acc[i][u] = fmaf(gpair[u], vv[j][u], acc[i][u]);
            acc[j][u] = fmaf(gpair[u], vv[i][u], acc[j][u]);
          }
          ++p;
        }
    float* o = gemb + b * ldg + e;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (i < nvec) {
        float* oi = o + (int64_t)i * dim;
        if (VEC == 4) {
          float4 t = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][VEC - 1]);
          if (accumulate) {
            const float4 old = *reinterpret_cast<const float4*>(oi);
            t = make_float4(old.x + t.x, old.y + t.y, old.z + t.z, old.w + t.w);
          }
          *reinterpret_cast<float4*>(oi) = t;
        } else {
          oi[0] = accumulate ? oi[0] + acc[i][0] : acc[i][0];
        }
      }
  }
}

}  // namespace

extern "C" int ctr_pairprod_fwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim, float* out,
                                int64_t ldo, void* stream) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(emb && out && nvec >= 2 && nvec <= 16 && dim > 0 && lde >= (int64_t)nvec * dim && ldo >= dim, CTR_EINVAL);
  const dim3 grid(ctr_stream_grid(batch * dim, kBiBlock));
  hipStream_t st = (hipStream_t)stream;
#define CTR_PP(NV_) hipLaunchKernelGGL(pairprod_fwd_kernel<NV_>, grid, dim3(kBiBlock), 0, st, emb, lde, batch, nvec, dim, out, ldo)
  switch (nvec) {
    case 2: CTR_PP(2); break;
    case 3: CTR_PP(3); break;
    case 4: CTR_PP(4); break;
    case 5: CTR_PP(5); break;
    case 6: CTR_PP(6); break;
    case 7: CTR_PP(7); break;
    case 8: CTR_PP(8); break;
    default: CTR_PP(16); break;
  }
#undef CTR_PP
  return ctr_launch_status();
}

extern "C" int ctr_pairprod_bwd(const float* emb, int64_t lde, int64_t batch, int nvec, int dim, const float* gp,
                                int64_t ldgp, const float* attn, const float* gpool, int64_t ldgo, float* gemb,
                                int64_t ldg, int accumulate, void* stream) {
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(emb && gp && gemb && nvec >= 2 && nvec <= 16 && dim > 0, CTR_EINVAL);
  CTR_REQUIRE(lde >= (int64_t)nvec * dim && ldg >= (int64_t)nvec * dim && ldgp >= dim, CTR_EINVAL);
  CTR_REQUIRE((attn == nullptr) == (gpool == nullptr) && (!gpool || ldgo >= dim), CTR_EINVAL);
  const bool vec4 = dim % 4 == 0 && lde % 4 == 0 && ldgp % 4 == 0 && ldg % 4 == 0 && ctr_aligned16(emb) &&
                    ctr_aligned16(gp) && ctr_aligned16(gemb) && (!gpool || (ldgo % 4 == 0 && ctr_aligned16(gpool)));
  const dim3 grid(ctr_stream_grid(batch * (dim / (vec4 ? 4 : 1)), kBiBlock));
  hipStream_t st = (hipStream_t)stream;
#define CTR_PP(NV_)                                                                                                    \
  do {                                                                                                                 \
    if (vec4)                                                                                                          \
      hipLaunchKernelGGL((pairprod_bwd_kernel<NV_, 4>), grid, dim3(kBiBlock), 0, st, emb, lde, batch, nvec, dim, gp, ldgp, \
                         attn, gpool, ldgo, gemb, ldg, accumulate);                                                    \
    else                                                                                                               \
      hipLaunchKernelGGL((pairprod_bwd_kernel<NV_, 1>), grid, dim3(kBiBlock), 0, st, emb, lde, batch, nvec, dim, gp, ldgp, \
                         attn, gpool, ldgo, gemb, ldg, accumulate);                                                    \
  } while (0)
  switch (nvec) {
    case 2: CTR_PP(2); break;
    case 3: CTR_PP(3); break;
    case 4: CTR_PP(4); break;
    case 5: CTR_PP(5); break;
    case 6: CTR_PP(6); break;
    case 7: CTR_PP(7); break;
    case 8: CTR_PP(8); break;
    default: CTR_PP(16); break;
  }
#undef CTR_PP
  return ctr_launch_status();
}
