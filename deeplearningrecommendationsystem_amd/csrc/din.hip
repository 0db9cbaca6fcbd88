// DIN / DIEN attention over the behaviour sequence (model/din.py:33-53,
// model/dien.py:23-39), the pieces around the attention MLP (which runs on the
// MFMA linear kernels):
//   concat_fwd : K3 sequence gather fused with the [h, h-t, t] operand build
//   pool_fwd   : softmax over L (no padding mask, as the reference) + weighted sum
//                (DIN) or per-position scaling (DIEN)
//   pool_bwd   : gradient of the scores through the softmax
//   concat_bwd : all gradient paths into the item table, scattered with fp32
//                atomics; the pad row (id 0) that every padded history hits is
//                pre-reduced per workgroup in LDS
// Lane mapping everywhere: LPR lanes share one (b,l) row (dwordx4 each when
// E % 4 == 0), a wave covers 64/LPR consecutive positions, coalesced in memory.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;

inline int pow2_ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

struct SeqGeom {
  int64_t batch;
  int len;   // L
  int dim;   // E
  int vec;   // 4 or 1
  int lpr;   // lanes per row (power of two <= 64)
};

inline SeqGeom make_geom(int64_t batch, int len, int dim, bool aligned) {
  SeqGeom g;
  g.batch = batch;
  g.len = len;
  g.dim = dim;
  g.vec = (aligned && dim % 4 == 0) ? 4 : 1;
  const int units = dim / g.vec;
  g.lpr = pow2_ceil(units);
  if (g.lpr > 64) g.lpr = 64;
  return g;
}

template <int VEC>
struct Pack;
template <>
struct Pack<1> {
  float v[1];
  __device__ __forceinline__ void load(const float* p) { v[0] = *p; }
  __device__ __forceinline__ void store(float* p) const { *p = v[0]; }
};
template <>
struct Pack<4> {
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};

__device__ __forceinline__ int64_t safe_row(int64_t r, int64_t vocab, int32_t* err) {
  if (r < 0 || r >= vocab) {
    if (err) *err = 1;
    return 0;
  }
  return r;
}

// triple: c[(b*L+l), 0:E] = h, [E:2E] = h - t, [2E:3E] = t ; pair: [0:E] = h, [E:2E] = t ;
// tvec[b, 0:E] = t
template <int VEC>
__global__ void __launch_bounds__(kBlock)
concat_fwd_kernel(const SeqGeom g, const float* __restrict__ table, int64_t vocab, const int64_t* __restrict__ hist,
                  const int64_t* __restrict__ target, float* __restrict__ c, int64_t ldc, float* __restrict__ tvec,
                  int64_t ldt, int32_t* err, int layout) {
  const int sub = threadIdx.x % g.lpr;
  const int64_t rows = g.batch * g.len;
  const int64_t groups = ((int64_t)gridDim.x * blockDim.x) / g.lpr;
  for (int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / g.lpr; row < rows; row += groups) {
    const int64_t b = row / g.len;
    const int l = (int)(row - b * g.len);
    const int64_t hr = safe_row(hist[row], vocab, err);
    const int64_t tr = safe_row(target[b], vocab, err);
    for (int e = sub * VEC; e < g.dim; e += g.lpr * VEC) {
      Pack<VEC> h, t, d;
      h.load(table + hr * g.dim + e);
      t.load(table + tr * g.dim + e);
#pragma unroll
      for (int v = 0; v < VEC; ++v) d.v[v] = h.v[v] - t.v[v];
      float* dst = c + row * ldc + e;
      h.store(dst);
      if (layout == CTR_DIN_PAIR) {
        t.store(dst + g.dim);
      } else if (layout == CTR_DIN_TRIPLE) {
        d.store(dst + g.dim);
        t.store(dst + 2 * g.dim);
      }  // CTR_DIN_H: the gathered rows only
      if (l == 0 && tvec) t.store(tvec + b * ldt + e);
    }
  }
}

// one wave per sample.  a = softmax_L(score); summed: out[b,:] = sum_l a_l h_l, else
// out[(b,l),:] = a_l h_l.  h_l is read from hsrc[(b*L+l)*ldh ...] (the first E columns
// of the concat operand).
template <int VEC>
__global__ void __launch_bounds__(kBlock)
pool_fwd_kernel(const SeqGeom g, const float* __restrict__ score, const float* __restrict__ hsrc, int64_t ldh,
                float* __restrict__ attn, float* __restrict__ out, int64_t ldo, int summed) {
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int sub = lane % g.lpr, rsub = lane / g.lpr, rows_per_iter = 64 / g.lpr;
  for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; b < g.batch; b += waves) {
    const float* s = score + b * g.len;
    float mx = -INFINITY;
    for (int l = lane; l < g.len; l += 64) mx = fmaxf(mx, s[l]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float den = 0.0f;
    for (int l = lane; l < g.len; l += 64) den += expf(s[l] - mx);
    den = ctr_wave_sum(den);

    float* ab = attn + b * g.len;
    for (int l = lane; l < g.len; l += 64) ab[l] = expf(s[l] - mx) / den;
    // weighted rows (a_l recomputed per row group: same expression, same value)
    float acc[8][VEC];
    const int chunks = (g.dim + g.lpr * VEC - 1) / (g.lpr * VEC);
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[q][v] = 0.0f;
    int l0 = 0;
    if (chunks == 1) {
      // a row fits one chunk per lane (E <= lpr * VEC, the BASELINE shapes): four row groups in flight --
      // one group at a time is a load-use loop that leaves the kernel latency-bound; same FMA order
      const int e = sub * VEC;
      const bool live = e < g.dim;
      for (; l0 < g.len; l0 += 4 * rows_per_iter) {  // every group guarded: the tail runs here too
        Pack<VEC> h4[4];
        float al4[4];
        bool ok4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int l = l0 + u * rows_per_iter + rsub;
          ok4[u] = live && l < g.len;
          const int lc = l < g.len ? l : g.len - 1;
          al4[u] = expf(s[lc] - mx) / den;
#pragma unroll
          for (int v = 0; v < VEC; ++v) h4[u].v[v] = 0.0f;
          if (ok4[u]) h4[u].load(hsrc + (b * g.len + l) * ldh + e);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!ok4[u]) continue;
          if (summed) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[0][v] = fmaf(al4[u], h4[u].v[v], acc[0][v]);
          } else {
            const int l = l0 + u * rows_per_iter + rsub;
            Pack<VEC> o;
#pragma unroll
            for (int v = 0; v < VEC; ++v) o.v[v] = h4[u].v[v] * al4[u];
            o.store(out + (b * g.len + l) * ldo + e);
          }
        }
      }
    }
    for (; l0 < g.len; l0 += rows_per_iter) {
      const int l = l0 + rsub;
      if (l < g.len) {
        const float al = expf(s[l] - mx) / den;
        const float* hr = hsrc + (b * g.len + l) * ldh;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int e = (sub + q * g.lpr) * VEC;
          if (q < chunks && e < g.dim) {
            Pack<VEC> h;
            h.load(hr + e);
            if (summed) {
#pragma unroll
              for (int v = 0; v < VEC; ++v) acc[q][v] = fmaf(al, h.v[v], acc[q][v]);
            } else {
              Pack<VEC> o;
#pragma unroll
              for (int v = 0; v < VEC; ++v) o.v[v] = h.v[v] * al;
              o.store(out + (b * g.len + l) * ldo + e);
            }
          }
        }
      }
    }
    if (summed) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = (sub + q * g.lpr) * VEC;
        if (q < chunks) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            float t = acc[q][v];
            for (int o = g.lpr; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
            acc[q][v] = t;
          }
          if (rsub == 0 && e < g.dim) {
            Pack<VEC> o;
#pragma unroll
            for (int v = 0; v < VEC; ++v) o.v[v] = acc[q][v];
            o.store(out + b * ldo + e);
          }
        }
      }
    }
  }
}

// gscore[b,l] = a_l (ga_l - sum_k a_k ga_k),  ga_l = <gsrc_l, h_l>, gsrc_l = gout[b,:]
// (summed) or gout[(b,l),:].  One sweep over the sample's rows when the ga_l of a sample fit the wave's strip of
// LDS (len <= kPoolStage: 320 -> us on 32768 x 100 x 64, the second sweep missed L2 with 2048 waves x 25 KB in
// flight); longer histories take a second sweep that recomputes ga_l bit-identically.
constexpr int kPoolStage = 512;
// sum over the 16 lanes of a DPP row, result in every lane: four rotate-and-add steps on the VALU.  The xor-shuffle
// form is a ds_bpermute per step -- 3.7 M of them per DIN backward through the LDS crossbar of 256 CUs.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, true));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, true));  // row_ror:1
  return v;
}
__device__ __forceinline__ float quad_sum4(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  return v;
}

template <int VEC>
__global__ void __launch_bounds__(kBlock)
pool_bwd_kernel(const SeqGeom g, const float* __restrict__ attn, const float* __restrict__ hsrc, int64_t ldh,
                const float* __restrict__ gout, int64_t ldgo, int summed, float* __restrict__ gscore) {
  __shared__ float s_ga[kBlock / 64][kPoolStage];
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int sub = lane % g.lpr, rsub = lane / g.lpr, rows_per_iter = 64 / g.lpr;
  const bool staged = g.len <= kPoolStage;
  float* mine = s_ga[threadIdx.x >> 6];
  for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; b < g.batch; b += waves) {
    const float* ab = attn + b * g.len;
    float* gs = gscore + b * g.len;
    float dotsum = 0.0f;
#pragma unroll 1
    for (int pass = 0; pass < (staged ? 1 : 2); ++pass) {
      int l0 = 0;
      if (g.dim <= g.lpr * VEC) {
        // single chunk per lane: four row groups in flight (see pool_fwd_kernel); per row the same dot product
        // and the same order of the dotsum / gscore updates
        const int e = sub * VEC;
        const bool live = e < g.dim;
        for (; l0 < g.len; l0 += 4 * rows_per_iter) {  // every group guarded: the tail runs here too
          float ga4[4], ab4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int l = l0 + u * rows_per_iter + rsub;
            ga4[u] = 0.0f;
            ab4[u] = ab[l < g.len ? l : g.len - 1];  // requested with the rows, not after the reduction
            if (live && l < g.len) {
              Pack<VEC> h, q;
              h.load(hsrc + (b * g.len + l) * ldh + e);
              q.load((summed ? gout + b * ldgo : gout + (b * g.len + l) * ldgo) + e);
#pragma unroll
              for (int v = 0; v < VEC; ++v) ga4[u] = fmaf(q.v[v], h.v[v], ga4[u]);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int l = l0 + u * rows_per_iter + rsub;
            float ga = ga4[u];
            if (g.lpr == 16) ga = row16_sum(ga);   // a row's lanes are one DPP row: VALU rotations, no LDS crossbar
            else if (g.lpr == 4) ga = quad_sum4(ga);
            else
              for (int o = g.lpr >> 1; o > 0; o >>= 1) ga += __shfl_xor(ga, o, 64);
            if (l < g.len && sub == 0) {
              if (pass == 0) {
                dotsum = fmaf(ab4[u], ga, dotsum);
                if (staged) mine[l] = ga;
              } else {
                gs[l] = ab4[u] * (ga - dotsum);
              }
            }
          }
        }
      }
      for (; l0 < g.len; l0 += rows_per_iter) {
        const int l = l0 + rsub;
        float ga = 0.0f;
        if (l < g.len) {
          const float* hr = hsrc + (b * g.len + l) * ldh;
          const float* gr = summed ? gout + b * ldgo : gout + (b * g.len + l) * ldgo;
          for (int e = sub * VEC; e < g.dim; e += g.lpr * VEC) {
            Pack<VEC> h, q;
            h.load(hr + e);
            q.load(gr + e);
#pragma unroll
            for (int v = 0; v < VEC; ++v) ga = fmaf(q.v[v], h.v[v], ga);
          }
        }
        for (int o = g.lpr >> 1; o > 0; o >>= 1) ga += __shfl_xor(ga, o, 64);
        if (l < g.len && sub == 0) {
          if (pass == 0) {
            dotsum = fmaf(ab[l], ga, dotsum);
            if (staged) mine[l] = ga;
          } else {
            gs[l] = ab[l] * (ga - dotsum);
          }
        }
      }
      if (pass == 0) dotsum = ctr_wave_sum(dotsum);
    }
    if (staged) {
      // the strip is the wave's own: its LDS writes are ordered before these reads, no workgroup barrier
      __builtin_amdgcn_wave_barrier();
      for (int l = lane; l < g.len; l += 64) gs[l] = ab[l] * (mine[l] - dotsum);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// Scatter every gradient path into the dense item-table gradient:
//   hist row  (b,l): g1 + g2 + a_l * gsrc_l        (gc = [g1,g2,g3] per position)
//   target row b   : sum_l (g3 - g2) + gt_extra[b]
template <int VEC>
__global__ void __launch_bounds__(kBlock)
concat_bwd_kernel(const SeqGeom g, const int64_t* __restrict__ hist, const int64_t* __restrict__ target,
                  int64_t vocab, const float* __restrict__ gc, int64_t ldc, const float* __restrict__ attn,
                  const float* __restrict__ gout, int64_t ldgo, int summed, const float* __restrict__ gt_extra,
                  int64_t ldgt, float* __restrict__ gtable, int pair) {
  extern __shared__ float s_pad[];  // E floats: gradient of row 0 accumulated by this workgroup
  for (int i = threadIdx.x; i < g.dim; i += blockDim.x) s_pad[i] = 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int sub = lane % g.lpr, rsub = lane / g.lpr, rows_per_iter = 64 / g.lpr;
  const int chunks = (g.dim + g.lpr * VEC - 1) / (g.lpr * VEC);
  for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; b < g.batch; b += waves) {
    float tacc[8][VEC];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int v = 0; v < VEC; ++v) tacc[q][v] = 0.0f;
    for (int l0 = 0; l0 < g.len; l0 += rows_per_iter) {
      const int l = l0 + rsub;
      if (l < g.len) {
        const int64_t row = b * g.len + l;
        int64_t hr = hist[row];
        if (hr < 0 || hr >= vocab) hr = 0;
        const float al = attn[row];
        const float* gr = summed ? gout + b * ldgo : gout + row * ldgo;
        const float* gcr = gc + row * ldc;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int e = (sub + q * g.lpr) * VEC;
          if (q < chunks && e < g.dim) {
            Pack<VEC> g1, g2, g3, go;
            g1.load(gcr + e);
            g2.load(gcr + g.dim + e);
            if (!pair) g3.load(gcr + 2 * g.dim + e);
            go.load(gr + e);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              // pair layout [gh, gt]: the h - t column was folded into the weights upstream
              const float gh = pair ? g1.v[v] + al * go.v[v] : (g1.v[v] + g2.v[v]) + al * go.v[v];
              tacc[q][v] += pair ? g2.v[v] : g3.v[v] - g2.v[v];
              if (hr == 0)
                atomicAdd(s_pad + e + v, gh);
              else
                unsafeAtomicAdd(gtable + hr * g.dim + e + v, gh);
            }
          }
        }
      }
    }
    int64_t tr = target[b];
    if (tr < 0 || tr >= vocab) tr = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = (sub + q * g.lpr) * VEC;
      if (q < chunks) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float t = tacc[q][v];
          for (int o = g.lpr; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
          if (rsub == 0 && e < g.dim) {
            if (gt_extra) t += gt_extra[b * ldgt + e + v];
            unsafeAtomicAdd(gtable + tr * g.dim + e + v, t);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < g.dim; i += blockDim.x) {
    const float v = s_pad[i];
    if (v != 0.0f) unsafeAtomicAdd(gtable + i, v);
  }
}

// Scatter of the history positions' gradient rows for the E-wide attention operand (CTR_DIN_H):
//   gtable[hist[b,l], e] += gh[(b,l), e] + attn[b,l] * gpool[b or (b,l), e]
// Flat (position, element) mapping, one dword per lane: the lanes of a wave cover whole 4*E-byte runs of ONE
// gradient row per atomic instruction (256 B at E = 64, the shape that runs at the memory-side atomic rate),
// UNROLL positions in flight per lane -- the wave-per-sample loop of concat_bwd_kernel is one dependent
// id -> row round trip per position (1.17 ms at cfg5 against a ~0.5 ms atomic floor).  Row 0 (the padding id, a
// quarter of all positions) is summed per workgroup in LDS.
template <int UNROLL>
__global__ void __launch_bounds__(kBlock)
seq_scatter_kernel(const int64_t* __restrict__ hist, int64_t vocab, uint32_t rows, int eshift, const CtrFastDiv div_len,
                   const float* __restrict__ gh, int64_t ldgh, const float* __restrict__ attn,
                   const float* __restrict__ gpool, int64_t ldgp, int summed, float* __restrict__ gtable) {
  extern __shared__ float s_pad[];
  const uint32_t dim = 1u << eshift;
  for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) s_pad[i] = 0.0f;
  __syncthreads();
  const uint32_t rpb = kBlock >> eshift;          // positions per workgroup pass and unroll slot
  const uint32_t e = threadIdx.x & (dim - 1), slot = threadIdx.x >> eshift;
  bool any0 = false;
  for (uint32_t base = blockIdx.x * rpb * UNROLL; base < rows; base += gridDim.x * rpb * UNROLL) {
    int64_t r[UNROLL];
    float v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      const uint32_t row = base + slot + k * rpb;
      r[k] = -1;
      v[k] = 0.0f;
      if (row < rows) {
        r[k] = ctr_ldg(hist + row);
        const uint32_t b = ctr_div(row, div_len);
        const float go = ctr_ldg(gpool + (summed ? (int64_t)b : (int64_t)row) * ldgp + e);
        v[k] = fmaf(ctr_ldg(attn + row), go, ctr_ldg(gh + (int64_t)row * ldgh + e));
      }
    }
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      if (r[k] > 0 && r[k] < vocab) ctr_atomic_add_global(gtable + (r[k] << eshift) + e, v[k]);
      else if (r[k] == 0) {
        atomicAdd(s_pad + e, v[k]);
        any0 = true;
      }
    }
  }
  if (__syncthreads_or(any0)) {
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) {
      const float t = s_pad[i];
      if (t != 0.0f) ctr_atomic_add_global(gtable + i, t);
    }
  }
}

inline int wave_grid(int64_t batch) {
  int64_t blocks = ctr_ceil_div(batch, kBlock / 64);
  return (int)(blocks < 2048 ? (blocks < 1 ? 1 : blocks) : 2048);
}

inline int check_dim(const SeqGeom& g) {
  // a lane owns at most 8 chunks of a row
  return g.lpr * g.vec * 8 >= g.dim ? CTR_OK : CTR_ELIMIT;
}

}  // namespace

extern "C" int ctr_din_concat_fwd(const float* table, int64_t vocab, int dim, const int64_t* hist,
                                  const int64_t* target, int64_t batch, int len, float* c, int64_t ldc, float* tvec,
                                  int64_t ldt, int layout, int32_t* err_flag, void* stream) {
  CTR_REQUIRE(batch >= 0 && len >= 0, CTR_EINVAL);
  if (batch == 0 || len == 0) return CTR_OK;
  CTR_REQUIRE(layout == CTR_DIN_TRIPLE || layout == CTR_DIN_PAIR || layout == CTR_DIN_H, CTR_EINVAL);
  const int width = layout == CTR_DIN_TRIPLE ? 3 : (layout == CTR_DIN_PAIR ? 2 : 1);
  CTR_REQUIRE(table && hist && target && c && vocab > 0 && dim > 0 && ldc >= width * (int64_t)dim, CTR_EINVAL);
  CTR_REQUIRE(!tvec || ldt >= dim, CTR_EINVAL);
  const bool al = ctr_aligned16(table) && ctr_aligned16(c) && ldc % 4 == 0 && (!tvec || (ctr_aligned16(tvec) && ldt % 4 == 0));
  const SeqGeom g = make_geom(batch, len, dim, al);
  const int grid = ctr_stream_grid(batch * len * g.lpr, kBlock);
  hipStream_t st = (hipStream_t)stream;
  if (g.vec == 4)
    hipLaunchKernelGGL(concat_fwd_kernel<4>, dim3(grid), dim3(kBlock), 0, st, g, table, vocab, hist, target, c, ldc,
                       tvec, ldt, err_flag, layout);
  else
    hipLaunchKernelGGL(concat_fwd_kernel<1>, dim3(grid), dim3(kBlock), 0, st, g, table, vocab, hist, target, c, ldc,
                       tvec, ldt, err_flag, layout);
  return ctr_launch_status();
}

extern "C" int ctr_din_pool_fwd(const float* score, const float* hsrc, int64_t ldh, int64_t batch, int len, int dim,
                                float* attn, float* out, int64_t ldo, int summed, void* stream) {
  CTR_REQUIRE(batch >= 0 && len >= 0, CTR_EINVAL);
  if (batch == 0 || len == 0) return CTR_OK;
  CTR_REQUIRE(score && hsrc && attn && out && dim > 0 && ldh >= dim && ldo >= dim, CTR_EINVAL);
  const bool al = ctr_aligned16(hsrc) && ldh % 4 == 0 && ctr_aligned16(out) && ldo % 4 == 0;
  const SeqGeom g = make_geom(batch, len, dim, al);
  int rc = check_dim(g);
  if (rc != CTR_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (g.vec == 4)
    hipLaunchKernelGGL(pool_fwd_kernel<4>, dim3(wave_grid(batch)), dim3(kBlock), 0, st, g, score, hsrc, ldh, attn, out,
                       ldo, summed);
  else
    hipLaunchKernelGGL(pool_fwd_kernel<1>, dim3(wave_grid(batch)), dim3(kBlock), 0, st, g, score, hsrc, ldh, attn, out,
                       ldo, summed);
  return ctr_launch_status();
}

extern "C" int ctr_din_pool_bwd(const float* attn, const float* hsrc, int64_t ldh, int64_t batch, int len, int dim,
                                const float* gout, int64_t ldgo, int summed, float* gscore, void* stream) {
  CTR_REQUIRE(batch >= 0 && len >= 0, CTR_EINVAL);
  if (batch == 0 || len == 0) return CTR_OK;
  CTR_REQUIRE(attn && hsrc && gout && gscore && dim > 0 && ldh >= dim && ldgo >= dim, CTR_EINVAL);
  const bool al = ctr_aligned16(hsrc) && ldh % 4 == 0 && ctr_aligned16(gout) && ldgo % 4 == 0;
  const SeqGeom g = make_geom(batch, len, dim, al);
  hipStream_t st = (hipStream_t)stream;
  if (g.vec == 4)
    hipLaunchKernelGGL(pool_bwd_kernel<4>, dim3(wave_grid(batch)), dim3(kBlock), 0, st, g, attn, hsrc, ldh, gout, ldgo,
                       summed, gscore);
  else
    hipLaunchKernelGGL(pool_bwd_kernel<1>, dim3(wave_grid(batch)), dim3(kBlock), 0, st, g, attn, hsrc, ldh, gout, ldgo,
                       summed, gscore);
  return ctr_launch_status();
}

extern "C" int ctr_din_concat_bwd(const int64_t* hist, const int64_t* target, int64_t vocab, int64_t batch, int len,
                                  int dim, const float* gc, int64_t ldc, const float* attn, const float* gout,
                                  int64_t ldgo, int summed, const float* gt_extra, int64_t ldgt, int layout,
                                  float* gtable, void* stream) {
  CTR_REQUIRE(batch >= 0 && len >= 0, CTR_EINVAL);
  if (batch == 0 || len == 0) return CTR_OK;
  CTR_REQUIRE(layout == CTR_DIN_TRIPLE || layout == CTR_DIN_PAIR, CTR_EINVAL);
  const int pair = layout == CTR_DIN_PAIR;
  CTR_REQUIRE(hist && target && gc && attn && gout && gtable && vocab > 0 && dim > 0, CTR_EINVAL);
  CTR_REQUIRE(ldc >= (pair ? 2 : 3) * (int64_t)dim && ldgo >= dim && (!gt_extra || ldgt >= dim), CTR_EINVAL);
  // one dword per lane: a wave's atomic instruction then adds to contiguous runs of a gradient
  // row (256 B for E >= 64), the shape that runs at the full memory-side atomic rate --
  // dwordx4 per lane strides the lanes 16 B apart and measured 3x slower on the row scatter
  const SeqGeom g = make_geom(batch, len, dim, false);
  int rc = check_dim(g);
  if (rc != CTR_OK) return rc;
  int grid = wave_grid(batch);
  if (grid > 1024) grid = 1024;
  hipStream_t st = (hipStream_t)stream;
  const size_t dyn = (size_t)dim * sizeof(float);
  if (g.vec == 4)
    hipLaunchKernelGGL(concat_bwd_kernel<4>, dim3(grid), dim3(kBlock), dyn, st, g, hist, target, vocab, gc, ldc, attn,
                       gout, ldgo, summed, gt_extra, ldgt, gtable, pair);
  else
    hipLaunchKernelGGL(concat_bwd_kernel<1>, dim3(grid), dim3(kBlock), dyn, st, g, hist, target, vocab, gc, ldc, attn,
                       gout, ldgo, summed, gt_extra, ldgt, gtable, pair);
  return ctr_launch_status();
}

extern "C" int ctr_din_scatter_bwd(const int64_t* hist, int64_t vocab, int64_t batch, int len, int dim, const float* gh,
                                   int64_t ldgh, const float* attn, const float* gpool, int64_t ldgp, int summed,
                                   float* gtable, void* stream) {
  CTR_REQUIRE(batch >= 0 && len >= 0, CTR_EINVAL);
  if (batch == 0 || len == 0) return CTR_OK;
  CTR_REQUIRE(hist && gh && attn && gpool && gtable && vocab > 0 && ldgh >= dim && ldgp >= dim, CTR_EINVAL);
  CTR_REQUIRE(dim >= 1 && dim <= 256 && (dim & (dim - 1)) == 0 && batch * len < (1ll << 32), CTR_ELIMIT);
  int eshift = 0;
  while ((1 << eshift) < dim) ++eshift;
  const uint32_t rows = (uint32_t)(batch * len);
  constexpr int kUnroll = 4;
  int64_t grid = ctr_ceil_div((int64_t)rows, (kBlock / dim) * kUnroll);
  if (grid > 256 * 8) grid = 256 * 8;
  hipLaunchKernelGGL(seq_scatter_kernel<kUnroll>, dim3((unsigned)grid), dim3(kBlock), (size_t)dim * sizeof(float),
                     (hipStream_t)stream, hist, vocab, rows, eshift, ctr_fastdiv((uint32_t)len), gh, ldgh, attn, gpool, ldgp,
                     summed, gtable);
  return ctr_launch_status();
}
