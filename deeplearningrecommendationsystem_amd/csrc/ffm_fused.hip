// FFM forward in ONE launch (model/ffm.py:46-86): the 12 field-aware vectors of a sample -- 4 id-row gathers
// (:56-59) and 8 "multi-hot matmul" bags (:48-55) -- are formed in registers, the 15 dot products (:62-80) and
// the logistic head (:82-86, with the reference's quirk: the cross scalar is added to every dense input before
// the linear layer) follow without the (B, 12k) operand ever being re-read; it is written once, for the backward.
// Round 1 ran embed_fwd (81 us) -> HBM -> ffm_head_fwd (59 us: LDS tiles of <= 64 samples, three barriers per
// tile) for 0.1 GB of traffic: latency-bound, 24 % of the HBM rate on its own algorithmic bytes.
//
// Mapping: LPR = k/4 lanes own one sample (a lane holds one dwordx4 of each of the 12 vectors: 48 registers), a
// wave covers 64/LPR samples; the sample's 45 feature columns are staged once per wave in a wave-private LDS
// strip (coalesced global load, broadcast LDS reads); sums over a vector's k elements are xor-shuffles inside the
// lane group.  Bag rows with a zero weight are skipped: fma(0, t, v) == v, so the in-order FMA chain over the
// remaining rows gives the same bits as the chain over all K rows (a one-hot slice still returns its row exactly).
// No MFMA, no workgroup barrier.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kCols = 45;      // (B,45) layout of data/reader.py:98-112
constexpr int kStrip = 48;     // floats per staged sample

struct FfmTables {
  const float* t[12];   // VECTORS order of model/ffm.py (age_user, age_item, gender_user, ... itemid_item)
  int64_t num_users, num_items;
  const float* user1;   // (num_users, 1)
  const float* item1;   // (num_items, 1)
  const float* w;       // (43) linear weight over x[:, 2:]
  const float* b;       // (1)
};

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float dot4(const ctr_f32x4& a, const ctr_f32x4& b) {
  float s = a.x * b.x;
  s = fmaf(a.y, b.y, s);
  s = fmaf(a.z, b.z, s);
  return fmaf(a.w, b.w, s);
}

__device__ __forceinline__ ctr_f32x4 ldrow(const float* p) { return *(const CTR_GLOBAL ctr_f32x4*)p; }

template <int LPR>
__global__ void __launch_bounds__(kBlock)
ffm_fused_fwd_kernel(const FfmTables T, const float* __restrict__ x, int64_t ldx, uint32_t batch,
                     float* __restrict__ emb, int64_t lde, float* __restrict__ prob, int64_t ldp, int32_t* err) {
  constexpr int K = LPR * 4;          // num_vector
  constexpr int SPW = 64 / LPR;       // samples per wave
  __shared__ float s_x[kBlock / 64][SPW * kStrip];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, slot = lane / LPR;
  float* xs = s_x[wave];
  const uint32_t wave_stride = gridDim.x * (kBlock / 64) * SPW;
  for (uint32_t b0 = (blockIdx.x * (kBlock / 64) + wave) * SPW; b0 < batch; b0 += wave_stride) {
    // stage the wave's SPW feature rows (coalesced), then every lane reads its sample's columns by broadcast
    for (int i = lane; i < SPW * kCols; i += 64) {
      const int s = i / kCols, c = i - s * kCols;
      xs[s * kStrip + c] = b0 + s < batch ? ctr_ldg(x + (int64_t)(b0 + s) * ldx + c) : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t b = b0 + slot;
    const bool live = b < batch;
    const float* xr = xs + slot * kStrip;
    int64_t u = (int64_t)xr[0], it = (int64_t)xr[1];
    if (u < 0 || u >= T.num_users || it < 0 || it >= T.num_items) {
      if (err && live) *err = 1;
      u = (u < 0 || u >= T.num_users) ? 0 : u;
      it = (it < 0 || it >= T.num_items) ? 0 : it;
    }
    ctr_f32x4 v[12];
    // id rows (ffm.py:56-59)
    v[8] = ldrow(T.t[8] + u * K + sub * 4);
    v[9] = ldrow(T.t[9] + u * K + sub * 4);
    v[10] = ldrow(T.t[10] + it * K + sub * 4);
    v[11] = ldrow(T.t[11] + it * K + sub * 4);
    const float bias_ids = ctr_ldg(T.user1 + u) + ctr_ldg(T.item1 + it);
    // bags (ffm.py:48-55): x[:, a:a+K] @ W, as an in-order FMA chain over the rows with a non-zero weight
    const ctr_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 8; ++f) v[f] = zero;
    auto bag = [&](int vu, int col0, int rows) {
      for (int j = 0; j < rows; ++j) {
        const float w = xr[col0 + j];
        if (w != 0.0f) {
          v[vu] += w * ldrow(T.t[vu] + j * K + sub * 4);          // contracted to FMAs per component
          v[vu + 1] += w * ldrow(T.t[vu + 1] + j * K + sub * 4);
        }
      }
    };
    bag(0, 2, 1);     // age: one row, real weight
    bag(2, 3, 2);     // gender one-hot
    bag(4, 5, 21);    // occupation one-hot
    bag(6, 26, 19);   // genre multi-hot
    if (live) {
#pragma unroll
      for (int f = 0; f < 12; ++f) *(CTR_GLOBAL ctr_f32x4*)(emb + (int64_t)b * lde + f * K + sub * 4) = v[f];
    }
    // the 15 field-aware dot products in the reference's order (ffm.py:62-80), summed left to right (:82)
    float cross;
#define CTR_FFM_DOT(A, B) group_sum<LPR>(dot4(v[A], v[B]))
    cross = CTR_FFM_DOT(0, 2);
    cross += CTR_FFM_DOT(0, 4);
    cross += CTR_FFM_DOT(1, 6);
    cross += CTR_FFM_DOT(0, 8);
    cross += CTR_FFM_DOT(1, 10);
    cross += CTR_FFM_DOT(2, 4);
    cross += CTR_FFM_DOT(3, 6);
    cross += CTR_FFM_DOT(2, 8);
    cross += CTR_FFM_DOT(3, 10);
    cross += CTR_FFM_DOT(5, 6);
    cross += CTR_FFM_DOT(4, 8);
    cross += CTR_FFM_DOT(5, 10);
    cross += CTR_FFM_DOT(6, 9);
    cross += CTR_FFM_DOT(7, 11);
    cross += CTR_FFM_DOT(9, 10);
#undef CTR_FFM_DOT
    // linear(x[:, 2:] + cross) (ffm.py:84-86): the 43 terms dealt to the lanes of the group
    float lin = 0.0f;
    for (int c = sub; c < kCols - 2; c += LPR) lin = fmaf(xr[2 + c] + cross, ctr_ldg(T.w + c), lin);
    lin = group_sum<LPR>(lin) + ctr_ldg(T.b);
    if (live && sub == 0) prob[(int64_t)b * ldp] = ctr_sigmoid(bias_ids + lin);
    __builtin_amdgcn_wave_barrier();  // the strip is rewritten by the next pass
  }
}

// Backward of the head and the 15 dots, same mapping (round 1: ffm_head_bwd_kernel on LDS tiles, 145 us):
//   dz = gprob * p (1 - p);  user1[u] += dz, item1[i] += dz;  lin_b += sum dz;  lin_w[c] += sum dz (x_c + cross)
//   gemb[b, f, :] = dz * sum(lin_w) * sum_{partners p of f} v_p        (d cross / d v_f, model/ffm.py:62-86)
// The vectors come back from the emb the forward wrote; the table gradients are then formed from gemb by the
// embedding backward (sorted segment sums for the id tables, register accumulation for the bag tables).
template <int LPR>
__global__ void __launch_bounds__(kBlock)
ffm_fused_bwd_kernel(const FfmTables T, const float* __restrict__ x, int64_t ldx, uint32_t batch,
                     const float* __restrict__ emb, int64_t lde, const float* __restrict__ prob, int64_t ldp,
                     const float* __restrict__ gprob, int64_t ldgp, float* __restrict__ guser1,
                     float* __restrict__ gitem1, float* __restrict__ gemb, int64_t ldg, float* __restrict__ ws) {
  constexpr int K = LPR * 4;
  constexpr int SPW = 64 / LPR;
  constexpr int NW = kCols - 2;                    // 43 dense weights
  constexpr int PER = (NW + LPR - 1) / LPR;        // weight slots per lane
  __shared__ float s_x[kBlock / 64][SPW * kStrip];
  __shared__ float s_acc[kBlock / 64][NW + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, slot = lane / LPR;
  float* xs = s_x[wave];
  float wsum = 0.0f;
  for (int c = 0; c < NW; ++c) wsum += ctr_ldg(T.w + c);   // same order in every lane
  float gw[PER];                                            // lane `sub`: columns sub, sub + LPR, ...
#pragma unroll
  for (int q = 0; q < PER; ++q) gw[q] = 0.0f;
  float gb = 0.0f;
  const uint32_t wave_stride = gridDim.x * (kBlock / 64) * SPW;
  for (uint32_t b0 = (blockIdx.x * (kBlock / 64) + wave) * SPW; b0 < batch; b0 += wave_stride) {
    for (int i = lane; i < SPW * kCols; i += 64) {
      const int s = i / kCols, c = i - s * kCols;
      xs[s * kStrip + c] = b0 + s < batch ? ctr_ldg(x + (int64_t)(b0 + s) * ldx + c) : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t b = b0 + slot;
    const bool live = b < batch;
    const uint32_t bb = live ? b : 0;
    const float* xr = xs + slot * kStrip;
    ctr_f32x4 v[12];
#pragma unroll
    for (int f = 0; f < 12; ++f) v[f] = ldrow(emb + (int64_t)bb * lde + f * K + sub * 4);
    const float p = ctr_ldg(prob + (int64_t)bb * ldp);
    const float dz = live ? ctr_ldg(gprob + (int64_t)bb * ldgp) * p * (1.0f - p) : 0.0f;
    float cross;
#define CTR_FFM_DOT(A, B) group_sum<LPR>(dot4(v[A], v[B]))
    cross = CTR_FFM_DOT(0, 2);
    cross += CTR_FFM_DOT(0, 4);
    cross += CTR_FFM_DOT(1, 6);
    cross += CTR_FFM_DOT(0, 8);
    cross += CTR_FFM_DOT(1, 10);
    cross += CTR_FFM_DOT(2, 4);
    cross += CTR_FFM_DOT(3, 6);
    cross += CTR_FFM_DOT(2, 8);
    cross += CTR_FFM_DOT(3, 10);
    cross += CTR_FFM_DOT(5, 6);
    cross += CTR_FFM_DOT(4, 8);
    cross += CTR_FFM_DOT(5, 10);
    cross += CTR_FFM_DOT(6, 9);
    cross += CTR_FFM_DOT(7, 11);
    cross += CTR_FFM_DOT(9, 10);
#undef CTR_FFM_DOT
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int c = sub + q * LPR;
      if (c < NW) gw[q] = fmaf(dz, xr[2 + c] + cross, gw[q]);
    }
    if (sub == 0) gb += dz;
    if (live && sub == 0) {
      int64_t u = (int64_t)xr[0], it = (int64_t)xr[1];
      if (guser1 && u >= 0 && u < T.num_users) ctr_atomic_add_global(guser1 + u, dz);
      if (gitem1 && it >= 0 && it < T.num_items) ctr_atomic_add_global(gitem1 + it, dz);
    }
    if (live && gemb) {
      const float dc = dz * wsum;
      float* g = gemb + (int64_t)b * ldg + sub * 4;
      // partners of each vector in the 15 pairs (ffm.py:62-80)
      const ctr_f32x4 a024 = v[2] + v[4] + v[8];        // partners of 0
      const ctr_f32x4 s610 = v[6] + v[10];              // partners of 1, 3, 5, 9
      const ctr_f32x4 s1359 = (v[1] + v[3]) + (v[5] + v[9]);   // partners of 6, 10
      *(CTR_GLOBAL ctr_f32x4*)(g + 0 * K) = dc * a024;
      *(CTR_GLOBAL ctr_f32x4*)(g + 1 * K) = dc * s610;
      *(CTR_GLOBAL ctr_f32x4*)(g + 2 * K) = dc * (v[0] + v[4] + v[8]);
      *(CTR_GLOBAL ctr_f32x4*)(g + 3 * K) = dc * s610;
      *(CTR_GLOBAL ctr_f32x4*)(g + 4 * K) = dc * (v[0] + v[2] + v[8]);
      *(CTR_GLOBAL ctr_f32x4*)(g + 5 * K) = dc * s610;
      *(CTR_GLOBAL ctr_f32x4*)(g + 6 * K) = dc * s1359;
      *(CTR_GLOBAL ctr_f32x4*)(g + 7 * K) = dc * v[11];
      *(CTR_GLOBAL ctr_f32x4*)(g + 8 * K) = dc * (v[0] + v[2] + v[4]);
      *(CTR_GLOBAL ctr_f32x4*)(g + 9 * K) = dc * s610;
      *(CTR_GLOBAL ctr_f32x4*)(g + 10 * K) = dc * s1359;
      *(CTR_GLOBAL ctr_f32x4*)(g + 11 * K) = dc * v[7];
    }
    __builtin_amdgcn_wave_barrier();
  }
  // weight / bias gradient partials: lanes of equal `sub` hold the same columns -> fold the sample slots of the
  // wave, then the waves, then one partial per workgroup to the workspace (summed in index order by reduce.hip)
#pragma unroll
  for (int q = 0; q < PER; ++q)
    for (int o = LPR; o < 64; o <<= 1) gw[q] += __shfl_xor(gw[q], o, 64);
  for (int o = LPR; o < 64; o <<= 1) gb += __shfl_xor(gb, o, 64);
  if (slot == 0) {
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (sub + q * LPR < NW) s_acc[wave][sub + q * LPR] = gw[q];
    if (sub == 0) s_acc[wave][NW] = gb;
  }
  __syncthreads();
  for (int c = threadIdx.x; c <= NW; c += blockDim.x) {
    float t = s_acc[0][c];
    for (int w2 = 1; w2 < kBlock / 64; ++w2) t += s_acc[w2][c];
    ws[(int64_t)blockIdx.x * (NW + 1) + c] = t;
  }
}

}  // namespace

extern "C" int ctr_ffm_fused_fwd(const float* x, int64_t ldx, int64_t batch, int dim, const float* const* tables,
                                 int64_t num_users, int64_t num_items, const float* user1, const float* item1,
                                 const float* lin_w, const float* lin_b, float* emb, int64_t lde, float* prob,
                                 int64_t ldp, int32_t* err_flag, void* stream) {
  CTR_REQUIRE(batch >= 0 && batch < (1ll << 31), CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(x && tables && user1 && item1 && lin_w && lin_b && emb && prob && ldx >= kCols && ldp >= 1, CTR_EINVAL);
  CTR_REQUIRE(num_users > 0 && num_items > 0 && lde >= 12 * (int64_t)dim, CTR_EINVAL);
  CTR_REQUIRE(dim == 8 || dim == 16 || dim == 32 || dim == 64, CTR_ELIMIT);
  CTR_REQUIRE(ctr_aligned16(emb) && lde % 4 == 0, CTR_EALIGN);
  FfmTables T;
  for (int f = 0; f < 12; ++f) {
    CTR_REQUIRE(tables[f], CTR_EINVAL);
    CTR_REQUIRE(ctr_aligned16(tables[f]), CTR_EALIGN);
    T.t[f] = tables[f];
  }
  T.num_users = num_users;
  T.num_items = num_items;
  T.user1 = user1;
  T.item1 = item1;
  T.w = lin_w;
  T.b = lin_b;
  const int lpr = dim / 4;
  const int spb = (kBlock / 64) * (64 / lpr);
  int64_t grid = ctr_ceil_div(batch, spb);
  if (grid > 256 * 8) grid = 256 * 8;
  hipStream_t st = (hipStream_t)stream;
#define CTR_FFM_FWD(L)                                                                                          \
  hipLaunchKernelGGL((ffm_fused_fwd_kernel<L>), dim3((unsigned)grid), dim3(kBlock), 0, st, T, x, ldx, (uint32_t)batch, \
                     emb, lde, prob, ldp, err_flag)
  switch (lpr) {
    case 2: CTR_FFM_FWD(2); break;
    case 4: CTR_FFM_FWD(4); break;
    case 8: CTR_FFM_FWD(8); break;
    default: CTR_FFM_FWD(16); break;
  }
#undef CTR_FFM_FWD
  return ctr_launch_status();
}

extern "C" int ctr_ffm_fused_bwd(const float* x, int64_t ldx, int64_t batch, int dim, const float* emb, int64_t lde,
                                 int64_t num_users, int64_t num_items, const float* lin_w, const float* prob,
                                 int64_t ldp, const float* gprob, int64_t ldgp, float* guser1, float* gitem1,
                                 float* glin_w, float* glin_b, float* gemb, int64_t ldg, float* workspace,
                                 int64_t workspace_floats, void* stream) {
  CTR_REQUIRE(batch >= 0 && batch < (1ll << 31), CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(x && emb && lin_w && prob && gprob && workspace && ldx >= kCols && ldp >= 1 && ldgp >= 1, CTR_EINVAL);
  CTR_REQUIRE(num_users > 0 && num_items > 0 && lde >= 12 * (int64_t)dim && (!gemb || ldg >= 12 * (int64_t)dim), CTR_EINVAL);
  CTR_REQUIRE(dim == 8 || dim == 16 || dim == 32 || dim == 64, CTR_ELIMIT);
  CTR_REQUIRE(ctr_aligned16(emb) && lde % 4 == 0 && (!gemb || (ctr_aligned16(gemb) && ldg % 4 == 0)), CTR_EALIGN);
  FfmTables T{};
  T.num_users = num_users;
  T.num_items = num_items;
  T.w = lin_w;
  const int lpr = dim / 4;
  const int spb = (kBlock / 64) * (64 / lpr);
  int64_t grid = ctr_ceil_div(batch, spb);
  if (grid > 1024) grid = 1024;
  const int nw = kCols - 2;
  CTR_REQUIRE(workspace_floats >= grid * (nw + 1), CTR_ELIMIT);
  hipStream_t st = (hipStream_t)stream;
#define CTR_FFM_BWD(L)                                                                                          \
  hipLaunchKernelGGL((ffm_fused_bwd_kernel<L>), dim3((unsigned)grid), dim3(kBlock), 0, st, T, x, ldx, (uint32_t)batch, \
                     emb, lde, prob, ldp, gprob, ldgp, guser1, gitem1, gemb, ldg, workspace)
  switch (lpr) {
    case 2: CTR_FFM_BWD(2); break;
    case 4: CTR_FFM_BWD(4); break;
    case 8: CTR_FFM_BWD(8); break;
    default: CTR_FFM_BWD(16); break;
  }
#undef CTR_FFM_BWD
  int rc = ctr_launch_status();
  if (rc != CTR_OK) return rc;
  CtrSegments segs;
  segs.n = 0;
  if (glin_w) segs.s[segs.n++] = CtrSegment{0, nw, glin_w};
  if (glin_b) segs.s[segs.n++] = CtrSegment{nw, 1, glin_b};
  if (segs.n == 0) return CTR_OK;
  return ctr_reduce_segments(workspace, (int)grid, nw + 1, segs, st);
}
