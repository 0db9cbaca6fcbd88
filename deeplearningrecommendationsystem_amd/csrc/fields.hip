// N single-id fields, fused: the embedding gather of every field AND the interaction that
// consumes the vectors, in one pass over the (B, F) index matrix.  This is what
// model/deepfm.py:45-54 + :63 + :71-77 become when every one of F columns is an id
// (BASELINE configs[2]: 26 fields x 1e6 rows x emb 16): the F vectors of a sample are fetched
// once, written to the (B, F*E) operand the deep MLP needs, and the FM second-order term plus the
// first-order sum leave as ONE scalar per sample -- no second kernel re-reading the operand.
// HBM-bound byte moving and wavefront reductions: no MFMA here by design.
//
// Forward mapping: E/4 lanes own one sample (a lane holds one dwordx4 of every vector), so a
// wave-instruction fetches 64/(E/4) independent rows and the sum over a vector's elements is a
// 2..4-step xor-shuffle inside the lane group.  Backward mapping: E lanes own one sample (one dword
// per lane), because a gradient row must be added with one dword per lane for the memory-side
// atomic units to see whole 4*E-byte row segments (MI355X_MICROARCH.md, Global float atomics).
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;

struct FieldTables {
  const float* table[CTR_MAX_FIELDS];
  const float* first[CTR_MAX_FIELDS];  // (vocab, 1) first-order weights, or NULL
  int64_t vocab[CTR_MAX_FIELDS];
};
struct FieldGrads {
  float* table[CTR_MAX_FIELDS];
  float* first[CTR_MAX_FIELDS];
  int64_t vocab[CTR_MAX_FIELDS];
};

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// fm[b] = (sum_f first_f[id_f] + bias) + 0.5 * sum_e[(sum_f v_fe)^2 - sum_f v_fe^2];  emb[b, f*E + e] = v_fe
//
// Latency, not bandwidth, is what a per-sample loop over the fields has to beat: a lane group that walks its F
// fields two at a time makes F/2 dependent id -> row round trips (101 us at 26 x 1e6 x 16, batch 65536, against 56
// us for the bare gather, which has every row of the batch in flight at once).  The loop therefore runs CHUNK
// fields at a time with the ids of the NEXT chunk requested before the rows of the current one, so the only
// dependent round trips left are ceil(F / CHUNK) row fetches.  (Fully unrolled over 32 fields the compiler
// hoists every load to the top: 300 registers, or 316 bytes of scratch under a 256-register cap.)
template <int LPR, int CHUNK>
__global__ void __launch_bounds__(kBlock)
fields_fm_fwd_kernel(const FieldTables T, int nfields, const int64_t* __restrict__ idx, int64_t ldidx, uint32_t batch,
                     const float* __restrict__ bias, float* __restrict__ emb, int64_t lde, float* __restrict__ fm,
                     int64_t ldfm, int32_t* err_flag) {
  constexpr int E = LPR * 4;
  __shared__ const float* s_tab[CTR_MAX_FIELDS + CHUNK];
  __shared__ const float* s_first[CTR_MAX_FIELDS + CHUNK];
  __shared__ int64_t s_vocab[CTR_MAX_FIELDS + CHUNK];
  if (threadIdx.x < CTR_MAX_FIELDS + CHUNK) {
    const bool in = (int)threadIdx.x < nfields;
    // fields past the end alias field 0 (their loads are issued and dropped: no branch in the chunk body)
    s_tab[threadIdx.x] = T.table[in ? threadIdx.x : 0];
    s_first[threadIdx.x] = in ? T.first[threadIdx.x] : nullptr;
    s_vocab[threadIdx.x] = T.vocab[in ? threadIdx.x : 0];
  }
  __syncthreads();
  const int sub = threadIdx.x % LPR;
  const uint32_t spb = kBlock / LPR;  // samples per workgroup pass
  const float b0 = bias ? bias[0] : 0.0f;
  const int last = nfields - 1;
  for (uint32_t base = blockIdx.x * spb; base < batch; base += gridDim.x * spb) {
    const uint32_t b = base + threadIdx.x / LPR;
    const bool live = b < batch;
    const uint32_t bb = live ? b : 0;
    const int64_t* irow = idx + (int64_t)bb * ldidx;
    float* erow = emb + (int64_t)bb * lde + sub * 4;
    ctr_f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    float lin = 0.0f;
    int64_t rn[CHUNK];
#pragma unroll
    for (int k = 0; k < CHUNK; ++k) rn[k] = ctr_ldg(irow + (k < last ? k : last));
#pragma unroll 1
    for (int c = 0; c < nfields; c += CHUNK) {
      int64_t r[CHUNK];
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) {
        r[k] = rn[k];
        if (r[k] < 0 || r[k] >= s_vocab[c + k]) {
          if (err_flag && c + k < nfields) *err_flag = 1;
          r[k] = 0;
        }
      }
      ctr_f32x4 v[CHUNK];
      float w1[CHUNK];
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) v[k] = *(const CTR_GLOBAL ctr_f32x4*)(s_tab[c + k] + r[k] * E + sub * 4);
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) {
        // lane `sub` of the group fetches the first-order weight of the fields f == sub (mod LPR)
        const float* fo = s_first[c + k];
        w1[k] = (fo != nullptr && ((c + k) % LPR) == sub) ? ctr_ldg(fo + r[k]) : 0.0f;
      }
      // ids of the next chunk: on their way while this chunk's rows arrive
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) {
        const int f = c + CHUNK + k;
        rn[k] = ctr_ldg(irow + (f < last ? f : last));
      }
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) {
        if (c + k < nfields) {
          if (live) *(CTR_GLOBAL ctr_f32x4*)(erow + (c + k) * E) = v[k];
          s += v[k];
          q += v[k] * v[k];
          lin += w1[k];
        }
      }
    }
    const ctr_f32x4 d = s * s - q;
    float part = (d.x + d.y) + (d.z + d.w);
    part = group_sum<LPR>(part);
    lin = group_sum<LPR>(lin);
    if (live && sub == 0) fm[(int64_t)b * ldfm] = (lin + b0) + 0.5f * part;
  }
}

// backward for g = gfm[b]:
//   gtable_f[id_f, e] += gdeep[b, f*E + e] + g * (S_e - v_fe),  S_e = sum_f v_fe  (v read back from emb)
//   gfirst_f[id_f] += g,   gbias += sum_b g
template <int E>
__global__ void __launch_bounds__(kBlock)
fields_fm_bwd_kernel(const FieldGrads G, int nfields, const int64_t* __restrict__ idx, int64_t ldidx, uint32_t batch,
                     const float* __restrict__ emb, int64_t lde, const float* __restrict__ gdeep, int64_t ldg,
                     const float* __restrict__ gfm, int64_t ldgfm, float* __restrict__ gbias_part) {
  __shared__ float* s_gtab[CTR_MAX_FIELDS];
  __shared__ float* s_gfirst[CTR_MAX_FIELDS];
  __shared__ int64_t s_vocab[CTR_MAX_FIELDS];
  __shared__ float s_bias[kBlock / 64];
  if (threadIdx.x < nfields) {
    s_gtab[threadIdx.x] = G.table[threadIdx.x];
    s_gfirst[threadIdx.x] = G.first[threadIdx.x];
    s_vocab[threadIdx.x] = G.vocab[threadIdx.x];
  }
  __syncthreads();
  const int e = threadIdx.x % E;
  const uint32_t spb = kBlock / E;
  float bsum = 0.0f;
  for (uint32_t base = blockIdx.x * spb; base < batch; base += gridDim.x * spb) {
    const uint32_t b = base + threadIdx.x / E;
    if (b >= batch) continue;
    const int64_t* irow = idx + (int64_t)b * ldidx;
    const float* vrow = emb + (int64_t)b * lde + e;
    const float g = gfm ? ctr_ldg(gfm + (int64_t)b * ldgfm) : 0.0f;
    if (e == 0) bsum += g;
    float S = 0.0f;
    if (gfm) {
      for (int f = 0; f < nfields; ++f) S += ctr_ldg(vrow + f * E);
    }
    const float* grow = gdeep ? gdeep + (int64_t)b * ldg + e : nullptr;
    for (int f = 0; f < nfields; ++f) {
      const int64_t r = ctr_ldg(irow + f);
      if (r < 0 || r >= s_vocab[f]) continue;  // bad id: flagged by the forward, no gradient
      float gv = grow ? ctr_ldg(grow + f * E) : 0.0f;
      if (gfm) gv = fmaf(g, S - ctr_ldg(vrow + f * E), gv);
      float* dst = s_gtab[f];
      if (dst) ctr_atomic_add_global(dst + r * E + e, gv);
      // first-order weight of field f: one lane of the sample's group adds g
      float* fo = s_gfirst[f];
      if (fo && (f % E) == e) ctr_atomic_add_global(fo + r, g);
    }
  }
  if (gbias_part) {
    bsum = ctr_wave_sum(bsum);
    if ((threadIdx.x & 63) == 0) s_bias[threadIdx.x >> 6] = bsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.0f;
      for (int w = 0; w < kBlock / 64; ++w) t += s_bias[w];
      gbias_part[blockIdx.x] = t;  // summed in index order by reduce.hip
    }
  }
}

int pack_fields(int nfields, int dim, const float* const* tables, const int64_t* vocabs) {
  CTR_REQUIRE(nfields >= 1 && nfields <= CTR_MAX_FIELDS && tables && vocabs, CTR_EINVAL);
  CTR_REQUIRE(dim == 8 || dim == 16 || dim == 32 || dim == 64, CTR_ELIMIT);
  for (int f = 0; f < nfields; ++f) {
    CTR_REQUIRE(tables[f] && vocabs[f] > 0 && vocabs[f] < (1ll << 31), CTR_EINVAL);
    CTR_REQUIRE(ctr_aligned16(tables[f]), CTR_EALIGN);
  }
  return CTR_OK;
}

}  // namespace

extern "C" int ctr_fields_fm_fwd(const int64_t* idx, int64_t ldidx, int64_t batch, int nfields, int dim,
                                 const float* const* tables, const int64_t* vocabs, const float* const* first,
                                 const float* bias, float* emb, int64_t lde, float* fm, int64_t ldfm,
                                 int32_t* err_flag, void* stream) {
  CTR_REQUIRE(batch >= 0 && batch < (1ll << 31), CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  int rc = pack_fields(nfields, dim, tables, vocabs);
  if (rc != CTR_OK) return rc;
  CTR_REQUIRE(idx && emb && fm && ldidx >= nfields && lde >= (int64_t)nfields * dim && ldfm >= 1, CTR_EINVAL);
  CTR_REQUIRE(ctr_aligned16(emb) && lde % 4 == 0, CTR_EALIGN);
  FieldTables T;
  for (int f = 0; f < nfields; ++f) {
    T.table[f] = tables[f];
    T.first[f] = first ? first[f] : nullptr;
    T.vocab[f] = vocabs[f];
  }
  hipStream_t st = (hipStream_t)stream;
  const int lpr = dim / 4;
  int64_t grid = ctr_ceil_div(batch, kBlock / lpr);
  if (grid > 256 * 16) grid = 256 * 16;
#define CTR_FIELDS_FWD(L)                                                                                             \
  hipLaunchKernelGGL((fields_fm_fwd_kernel<L, 8>), dim3((unsigned)grid), dim3(kBlock), 0, st, T, nfields, idx, ldidx, \
                     (uint32_t)batch, bias, emb, lde, fm, ldfm, err_flag)
  switch (lpr) {
    case 2: CTR_FIELDS_FWD(2); break;
    case 4: CTR_FIELDS_FWD(4); break;
    case 8: CTR_FIELDS_FWD(8); break;
    default: CTR_FIELDS_FWD(16); break;
  }
#undef CTR_FIELDS_FWD
  return ctr_launch_status();
}

extern "C" int ctr_fields_fm_bwd(const int64_t* idx, int64_t ldidx, int64_t batch, int nfields, int dim,
                                 const int64_t* vocabs, const float* emb, int64_t lde, const float* gdeep, int64_t ldg,
                                 const float* gfm, int64_t ldgfm, float* const* gtables, float* const* gfirst,
                                 float* gbias, float* workspace, int64_t workspace_floats, void* stream) {
  CTR_REQUIRE(batch >= 0 && batch < (1ll << 31), CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(nfields >= 1 && nfields <= CTR_MAX_FIELDS && vocabs && gtables, CTR_EINVAL);
  CTR_REQUIRE(dim == 8 || dim == 16 || dim == 32 || dim == 64, CTR_ELIMIT);
  CTR_REQUIRE(idx && emb && ldidx >= nfields && lde >= (int64_t)nfields * dim, CTR_EINVAL);
  CTR_REQUIRE((gdeep == nullptr || ldg >= (int64_t)nfields * dim) && (gfm == nullptr || ldgfm >= 1), CTR_EINVAL);
  CTR_REQUIRE(gdeep || gfm, CTR_EINVAL);
  FieldGrads G;
  for (int f = 0; f < nfields; ++f) {
    CTR_REQUIRE(vocabs[f] > 0, CTR_EINVAL);
    G.table[f] = gtables[f];
    G.first[f] = (gfirst && gfm) ? gfirst[f] : nullptr;
    G.vocab[f] = vocabs[f];
  }
  hipStream_t st = (hipStream_t)stream;
  int64_t grid = ctr_ceil_div(batch, kBlock / dim);
  if (grid > 256 * 8) grid = 256 * 8;
  float* part = nullptr;
  if (gbias && gfm) {
    CTR_REQUIRE(workspace && workspace_floats >= grid, CTR_ELIMIT);
    part = workspace;
  }
#define CTR_FIELDS_BWD(D)                                                                                            \
  hipLaunchKernelGGL((fields_fm_bwd_kernel<D>), dim3((unsigned)grid), dim3(kBlock), 0, st, G, nfields, idx, ldidx,   \
                     (uint32_t)batch, emb, lde, gdeep, ldg, gfm, ldgfm, part)
  switch (dim) {
    case 8: CTR_FIELDS_BWD(8); break;
    case 16: CTR_FIELDS_BWD(16); break;
    case 32: CTR_FIELDS_BWD(32); break;
    default: CTR_FIELDS_BWD(64); break;
  }
#undef CTR_FIELDS_BWD
  int rc = ctr_launch_status();
  if (rc != CTR_OK || !part) return rc;
  CtrSegments segs;
  segs.n = 1;
  segs.s[0] = CtrSegment{0, 1, gbias};
  return ctr_reduce_segments(part, (int)grid, 1, segs, st);
}
