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

#include <type_traits>
#include <utility>

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

// ---------------------------------------------------------------------------------------------------
// All-pairs inner products for many fields (PNN, model/pnn.py:59-66, at F = 26: 325 products per sample).
// The six-field kernels of interact.hip stage LDS tiles of <= 64 samples with workgroup barriers; at 26 vectors a
// tile is 14..23 samples and they run at 18 % of the HBM rate (132 / 275 us).  Here LPR = E/4 lanes own a sample
// and a wave's 64/LPR samples sit in a wave-private LDS strip (coalesced copy in, no workgroup barrier): a lane
// keeps vector i in registers while it walks j, every product is 4 FMAs + an xor-shuffle sum over the lane group.
// Strip stride per sample = F*E + pad with (F*E + pad) % 64 == 16 floats: the 16 lanes of a ds_read_b128 group
// (4 samples x 4 lanes) then cover all 64 banks.
constexpr int kPairsMaxWaves = 2;   // waves per workgroup: as many (<= 2) as keep a workgroup's strips under 56 KB

__host__ __device__ constexpr int pairs_stride(int fe) { return fe + ((16 - fe % 64) + 64) % 64; }

template <int LPR>
__global__ void __launch_bounds__(64 * kPairsMaxWaves)
fields_pairs_fwd_kernel(const float* __restrict__ emb, int64_t lde, uint32_t batch, int nf, float* __restrict__ prod,
                        int64_t ldp) {
  constexpr int E = LPR * 4, SPW = 64 / LPR;
  extern __shared__ __attribute__((aligned(16))) float s_v[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, slot = lane / LPR;
  const int fe = nf * E, stride = pairs_stride(fe);
  float* strip = s_v + wave * SPW * stride;
  const int npairs = nf * (nf - 1) / 2;
  const uint32_t wpb = blockDim.x >> 6;
  const uint32_t wave_stride = gridDim.x * wpb * SPW;
  for (uint32_t b0 = (blockIdx.x * wpb + wave) * SPW; b0 < batch; b0 += wave_stride) {
    // copy the wave's samples in: dwordx4 units, consecutive lanes on consecutive units of a row
    const int upr = fe / 4;
    for (int i = lane; i < SPW * upr; i += 64) {
      const int smp = i / upr, u = i - smp * upr;
      ctr_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (b0 + smp < batch) v = *(const CTR_GLOBAL ctr_f32x4*)(emb + (int64_t)(b0 + smp) * lde + u * 4);
      *(ctr_f32x4*)(strip + smp * stride + u * 4) = v;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t b = b0 + slot;
    const float* mine = strip + slot * stride;
    float* out = prod + (int64_t)(b < batch ? b : 0) * ldp;
    // lane `sub` of the sample's group takes the pairs (i, j) of row i with (j - i - 1) % LPR == sub: a whole
    // dot product per lane (E FMAs, no cross-lane sum: a shuffle is an LDS round trip on this hardware), the
    // group's LPR lanes store LPR consecutive products; two rows of vector reads in flight per step
    for (int i = 0; i + 1 < nf; ++i) {
      ctr_f32x4 vi[LPR];
#pragma unroll
      for (int q = 0; q < LPR; ++q) vi[q] = *(const ctr_f32x4*)(mine + i * E + q * 4);
      const int row_p = i * nf - i * (i + 1) / 2;   // index of pair (i, i + 1)
      for (int j0 = i + 1 + sub; j0 < nf; j0 += 2 * LPR) {
        const int j1 = j0 + LPR;
        const bool two = j1 < nf;
        const float* pj0 = mine + j0 * E;
        const float* pj1 = mine + (two ? j1 : j0) * E;
        float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
        for (int q = 0; q < LPR; ++q) {
          const ctr_f32x4 a = *(const ctr_f32x4*)(pj0 + q * 4), c = *(const ctr_f32x4*)(pj1 + q * 4);
          d0 = fmaf(vi[q].x, a.x, d0); d0 = fmaf(vi[q].y, a.y, d0); d0 = fmaf(vi[q].z, a.z, d0); d0 = fmaf(vi[q].w, a.w, d0);
          d1 = fmaf(vi[q].x, c.x, d1); d1 = fmaf(vi[q].y, c.y, d1); d1 = fmaf(vi[q].z, c.z, d1); d1 = fmaf(vi[q].w, c.w, d1);
        }
        if (b < batch) {
          out[row_p + (j0 - i - 1)] = d0;
          if (two) out[row_p + (j1 - i - 1)] = d1;
        }
      }
    }
    (void)npairs;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- pinned shape: F vectors of E = 16 entirely in registers (BASELINE configs[2]: F = 26, 325 products).
// A lane group of 4 owns a sample, a lane holds one dwordx4 of every vector (F x 4 registers); the pair loop is
// fully unrolled, so every register index is a compile-time constant.  Forward: 4 FMAs per product and a
// quad-permute DPP sum (VALU, not an LDS round trip like ds_bpermute); lane q of the group collects products
// 16m + 4q .. +3 and stores them as one dwordx4.  Backward: both updates of a pair (acc_i += c v_j, acc_j += c v_i)
// on registers, the coefficients fetched as dwordx4 (all four lanes of a group read the same 16 bytes).
__host__ __device__ constexpr int pair_row(int p, int nf) {   // i of the p-th pair (i < j, lexicographic)
  int i = 0, rem = p;
  while (rem >= nf - 1 - i) { rem -= nf - 1 - i; ++i; }
  return i;
}
__host__ __device__ constexpr int pair_col(int p, int nf) {
  int i = 0, rem = p;
  while (rem >= nf - 1 - i) { rem -= nf - 1 - i; ++i; }
  return i + 1 + rem;
}

__device__ __forceinline__ float quad_sum(float v) {
  // lanes 4k .. 4k+3: v + xor-1 neighbour, then + xor-2 neighbour (DPP quad_perm [1,0,3,2] / [2,3,0,1])
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  return v;
}

template <int F, int P0, int N>   // products P0 .. P0+N-1 of the unrolled pair list
struct PairBlock {
  static __device__ __forceinline__ void fwd(const ctr_f32x4 (&v)[F], int sub, ctr_f32x4& o) {
    if constexpr (N > 0) {
      constexpr int p = P0, i = pair_row(p, F), j = pair_col(p, F);
      const ctr_f32x4 t = v[i] * v[j];
      const float d = quad_sum((t.x + t.y) + (t.z + t.w));
      constexpr int owner = (p >> 2) & 3, k = p & 3;
      if (sub == owner) o[k] = d;
      PairBlock<F, P0 + 1, N - 1>::fwd(v, sub, o);
    }
  }
  // backward: lane q of a quad holds the coefficients of products 16g + 4q .. +3 (one coalesced 64-byte read per
  // quad); the owner's value reaches the other three lanes through a quad_perm [q,q,q,q] DPP move
  static __device__ __forceinline__ void bwd(const ctr_f32x4 (&v)[F], ctr_f32x4 (&acc)[F], const ctr_f32x4& c) {
    if constexpr (N > 0) {
      constexpr int p = P0, i = pair_row(p, F), j = pair_col(p, F);
      constexpr int owner = (p >> 2) & 3, k = p & 3;
      const float mine = c[k];  // by value first: __builtin_bit_cast on the element lvalue c[k] reads element 0
      const float cc = __builtin_bit_cast(
          float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine), owner * 0x55, 0xF, 0xF, true));
      acc[i] += cc * v[j];
      acc[j] += cc * v[i];
      PairBlock<F, P0 + 1, N - 1>::bwd(v, acc, c);
    }
  }
};

template <class Fn, int... I>
__device__ __forceinline__ void static_for(Fn& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

template <int F>
__global__ void __launch_bounds__(kBlock)
pairs_reg_fwd_kernel(const float* __restrict__ emb, int64_t lde, uint32_t batch, float* __restrict__ prod, int64_t ldp) {
  constexpr int E = 16, P = F * (F - 1) / 2, NB = (P + 15) / 16;
  const int sub = threadIdx.x & 3;
  const uint32_t spb = kBlock / 4;
  for (uint32_t base = blockIdx.x * spb; base < batch; base += gridDim.x * spb) {
    const uint32_t b = base + (threadIdx.x >> 2);
    const bool live = b < batch;
    const float* row = emb + (int64_t)(live ? b : 0) * lde + sub * 4;
    ctr_f32x4 v[F];
#pragma unroll
    for (int f = 0; f < F; ++f) v[f] = *(const CTR_GLOBAL ctr_f32x4*)(row + f * E);
    float* out = prod + (int64_t)(live ? b : 0) * ldp;
    auto block = [&](auto mv) __attribute__((always_inline)) {
      constexpr int m = decltype(mv)::value;
      constexpr int n = P - 16 * m < 16 ? P - 16 * m : 16;
      ctr_f32x4 o = {0.f, 0.f, 0.f, 0.f};
      PairBlock<F, 16 * m, n>::fwd(v, sub, o);
      if (live) {
        if constexpr (n == 16) {
          *(CTR_GLOBAL ctr_f32x4*)(out + 16 * m + 4 * sub) = o;   // ldp % 4 == 0, out 16-byte aligned (host)
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (16 * m + 4 * sub + k < P) out[16 * m + 4 * sub + k] = o[k];
        }
      }
    };
    static_for(block, std::make_integer_sequence<int, NB>{});
  }
}

// Keeps the updates of one coefficient group in front of the next group's: the FMAs touch no memory, so nothing else
// stops the compiler from sinking all of them behind the last group with every broadcast value live (it did: 512
// registers + scratch).  An empty volatile asm that "modifies" the accumulators is an ordering point for them.
template <int F>
__device__ __forceinline__ void pin_accumulators(ctr_f32x4 (&a)[F]) {
  static_assert(F == 26, "written out for the pinned field count");
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                    "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]));
  asm volatile("" : "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]),
                    "+v"(a[20]), "+v"(a[21]), "+v"(a[22]), "+v"(a[23]), "+v"(a[24]), "+v"(a[25]));
}

// 316 VGPR + 60 AGPR: one wave per SIMD, every read of a sample in flight at once.  Capped at 256 registers (two waves
// per SIMD, 244 B of scratch) it measured 121.7 us against 102.2 us at 65536 x 26 x 16.
template <int F>
__global__ void __launch_bounds__(kBlock)
pairs_reg_bwd_kernel(const float* __restrict__ emb, int64_t lde, uint32_t batch, const float* __restrict__ gp,
                     int64_t ldgp, float* __restrict__ gemb, int64_t ldg, int accumulate) {
  constexpr int E = 16, P = F * (F - 1) / 2, NC = (P + 15) / 16;
  const int sub = threadIdx.x & 3;
  const uint32_t spb = kBlock / 4;
  for (uint32_t base = blockIdx.x * spb; base < batch; base += gridDim.x * spb) {
    const uint32_t b = base + (threadIdx.x >> 2);
    if (b >= batch) continue;  // a quad shares b: DPP partners are all live or all gone; nothing below is divergent
    const float* row = emb + (int64_t)b * lde + sub * 4;
    const float* cf = gp + (int64_t)b * ldgp;
    ctr_f32x4 v[F], acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      v[f] = *(const CTR_GLOBAL ctr_f32x4*)(row + f * E);
      acc[f] = ctr_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // every coefficient read is issued up front (NC x 4 registers); the groups then run in order -- without the
    // scheduling barrier the compiler hoists all F(F-1)/2 broadcasts and spills
    ctr_f32x4 cg[NC];
    auto fetch = [&](auto gv) __attribute__((always_inline)) {
      constexpr int g = decltype(gv)::value;
      // the padded row (ldgp % 4 == 0) keeps a partly used 16-byte read legal; lanes wholly past the row read nothing
      cg[g] = ctr_f32x4{0.f, 0.f, 0.f, 0.f};
      if (16 * g + 16 <= P || 16 * g + 4 * sub < P) cg[g] = *(const CTR_GLOBAL ctr_f32x4*)(cf + 16 * g + 4 * sub);
    };
    static_for(fetch, std::make_integer_sequence<int, NC>{});
    auto chunk = [&](auto gv) __attribute__((always_inline)) {
      constexpr int g = decltype(gv)::value;
      constexpr int n = P - 16 * g < 16 ? P - 16 * g : 16;
      PairBlock<F, 16 * g, n>::bwd(v, acc, cg[g]);
      pin_accumulators(acc);
    };
    static_for(chunk, std::make_integer_sequence<int, NC>{});
    float* dst = gemb + (int64_t)b * ldg + sub * 4;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      ctr_f32x4 r = acc[f];
      if (accumulate) r += *(const CTR_GLOBAL ctr_f32x4*)(dst + f * E);
      *(CTR_GLOBAL ctr_f32x4*)(dst + f * E) = r;
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

static int pairs_geometry(int64_t batch, int nfields, int dim, int extra, size_t* lds_bytes, int64_t* grid, int* waves) {
  CTR_REQUIRE(nfields >= 2 && nfields <= 64, CTR_ELIMIT);
  CTR_REQUIRE(dim == 8 || dim == 16 || dim == 32 || dim == 64, CTR_ELIMIT);
  const int spw = 64 / (dim / 4);
  const size_t per_wave = sizeof(float) * (size_t)spw * pairs_stride(nfields * dim + extra);
  CTR_REQUIRE(per_wave <= 56 * 1024, CTR_ELIMIT);
  const int w = 2 * per_wave <= 56 * 1024 ? 2 : 1;
  *waves = w;
  *lds_bytes = per_wave * w;
  int64_t g = ctr_ceil_div(batch, w * spw);
  if (g > 256 * 16) g = 256 * 16;
  *grid = g;
  return CTR_OK;
}

// same contract as ctr_allpairs_fwd / ctr_allpairs_bwd (include/ctrhip.h) for many vectors
extern "C" int ctr_fields_pairs_fwd(const float* emb, int64_t lde, int64_t batch, int nfields, int dim, float* prod,
                                    int64_t ldp, void* stream) {
  CTR_REQUIRE(batch >= 0 && batch < (1ll << 31), CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  const int npairs = nfields * (nfields - 1) / 2;
  CTR_REQUIRE(emb && prod && lde >= (int64_t)nfields * dim && ldp >= npairs, CTR_EINVAL);
  CTR_REQUIRE(ctr_aligned16(emb) && lde % 4 == 0, CTR_EALIGN);
  if (nfields == 26 && dim == 16 && ctr_aligned16(prod) && ldp % 4 == 0) {  // the pinned BASELINE shape
    int64_t g = ctr_ceil_div(batch, kBlock / 4);
    if (g > 256 * 8) g = 256 * 8;
    hipLaunchKernelGGL((pairs_reg_fwd_kernel<26>), dim3((unsigned)g), dim3(kBlock), 0, (hipStream_t)stream, emb, lde,
                       (uint32_t)batch, prod, ldp);
    return ctr_launch_status();
  }
  size_t lds;
  int64_t grid;
  int waves;
  int rc = pairs_geometry(batch, nfields, dim, 0, &lds, &grid, &waves);
  if (rc != CTR_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
#define CTR_PAIRS_FWD(L)                                                                                              \
  do {                                                                                                                \
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fields_pairs_fwd_kernel<L>),             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)   \
      return CTR_ELAUNCH;                                                                                             \
    hipLaunchKernelGGL((fields_pairs_fwd_kernel<L>), dim3((unsigned)grid), dim3(64 * waves), lds, st, emb, \
                       lde, (uint32_t)batch, nfields, prod, ldp);                                                     \
  } while (0)
  switch (dim / 4) {
    case 2: CTR_PAIRS_FWD(2); break;
    case 4: CTR_PAIRS_FWD(4); break;
    case 8: CTR_PAIRS_FWD(8); break;
    default: CTR_PAIRS_FWD(16); break;
  }
#undef CTR_PAIRS_FWD
  return ctr_launch_status();
}

extern "C" int ctr_fields_pairs_bwd(const float* emb, int64_t lde, int64_t batch, int nfields, int dim, const float* gp,
                                    int64_t ldgp, float* gemb, int64_t ldg, int accumulate, void* stream) {
  CTR_REQUIRE(batch >= 0 && batch < (1ll << 31), CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  const int npairs = nfields * (nfields - 1) / 2;
  CTR_REQUIRE(emb && gp && gemb && lde >= (int64_t)nfields * dim && ldg >= (int64_t)nfields * dim && ldgp >= npairs,
              CTR_EINVAL);
  CTR_REQUIRE(ctr_aligned16(emb) && lde % 4 == 0 && ctr_aligned16(gemb) && ldg % 4 == 0, CTR_EALIGN);
  if (nfields == 26 && dim == 16 && ctr_aligned16(gp) && ldgp % 4 == 0) {  // the pinned BASELINE shape
    int64_t g = ctr_ceil_div(batch, kBlock / 4);
    if (g > 256 * 8) g = 256 * 8;
    hipLaunchKernelGGL((pairs_reg_bwd_kernel<26>), dim3((unsigned)g), dim3(kBlock), 0, (hipStream_t)stream, emb, lde,
                       (uint32_t)batch, gp, ldgp, gemb, ldg, accumulate);
    return ctr_launch_status();
  }
  // other shapes: refused, the caller takes ctr_allpairs_bwd (an LDS-strip backward in the style of the forward
  // measured 692 us against that kernel's 275 us at 65536 x 26 x 16 and was dropped)
  return CTR_ELIMIT;
}
