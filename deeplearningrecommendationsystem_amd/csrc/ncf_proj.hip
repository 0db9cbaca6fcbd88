// NeuralCF when the vocabularies are much smaller than the batch (BASELINE configs[1]: 943 users / 1682 items, batch
// 65536): the first tower layer is moved from the SAMPLES to the TABLE ROWS.
//
// The reference computes  z0[b] = W0 . cat(MLP_U[u_b], MLP_I[i_b]) + b0  (model/neuralcf.py:43-49) for every sample:
// 2 * 128 * 64 flops per sample, three quarters of the tower's arithmetic, and its backward another 2x that plus a
// (B, 128) input gradient that the embedding backward has to segment-sum by row.  But a linear layer on a
// concatenation of two gathered rows is the sum of two gathered PROJECTED rows:
//     z0[b] = P_U[u_b] + P_I[i_b],      P_U = MLP_U . W0[:, :64]^T  (U x 64),   P_I = MLP_I . W0[:, 64:]^T + b0  (I x 64)
// -- two small matrix products over U + I = 2625 rows instead of one over 65536 samples -- and the chain rule gives
//     S_U[u] = sum_{b: u_b = u} gz0[b]   (same for S_I),        gz0 = relu'(z0) * (W1^T gz1)
//     dMLP_U = S_U . W0[:, :64],   dW0[:, :64] = S_U^T . MLP_U,   db0 = column sums of S_U
// so layer 0's dX and dW GEMMs over the batch (128 + 128 of the 344 matrix instructions a sample group cost the
// per-sample kernel, mlp_mfma16.hip) become products over the table rows as well.  The GMF half folds the same way:
//     T_U[u] = sum_{b: u_b = u} gz_b * GMF_I[i_b],   dGMF_U = wfold[:64] * T_U,   gwfold[:64] = sum_u GMF_U[u] * T_U[u]
// (gz_b = the head's pre-activation gradient, a scalar per sample).  What is left per sample is the 64-32-16-8 tower,
// the head's dot product and ONE 64-float row gz0[b] -- which is all the segment sums need.
//
// The segment sums are formed without a sort: the forward's id pass counts the samples of every row with returning
// atomics (rank of a sample within its row), the backward kernel scans the counts (every workgroup for itself, 2625
// entries) and stores each sample's gz0 row straight into its row's bucket, slot = offset[row] + rank, once for the
// user and once for the item; a streaming kernel then sums the buckets (balanced over SLOT ranges, so a hot row of a
// skewed id distribution is shared by many waves).  Same values as the per-sample path up to fp32 summation order;
// the order inside a bucket follows the atomics, so table gradients are reproducible to rounding, not bitwise.
//
// Launches: forward  ncfp_prep (projected tables, head fold) -> ncfp_fwd;   backward  ncfp_bwd -> reduce_segments_fold
// (tower dW partials, head fold chain rule) -> ncfp_segsum -> ncfp_finish (the table-row products).
#include "ctr_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kL = 3;                          // the layers that stay per sample
constexpr int kK[kL] = {64, 32, 16};
constexpr int kN[kL] = {32, 16, 8};
constexpr int kH = 64;                         // width of an MLP embedding row = half of layer 0's input
constexpr int kN0 = 64;                        // layer-0 units = width of a projected row
constexpr int kP = 64;                         // GMF width = head's extra columns
constexpr int kNL = 8;                         // head: last activations
constexpr int kHeadW = kP + kNL;

#include "mfma16_tower.inc"

struct Ids {
  const int64_t* uidx; int64_t ustride;
  const int64_t* iidx; int64_t istride;
  int64_t nu, ni;
};

// ------------------------------------------------------------------ prep: projected tables + head fold
struct Prep {
  const float* mlp_u; const float* mlp_i;      // (nu, 64), (ni, 64)
  const float* w0; int64_t ldw0; const float* b0;   // layer 0: (64, 128), (64)
  float* ptab;                                 // (nu + ni, 64): P_U rows, then P_I rows
  int64_t nu, ni;
  int32_t* counts; int64_t ncounts;            // nullable: zeroed (the forward's per-row sample counters)
  // head fold (ctr_fold_head_fwd's map for p = 64, n = 64, k = 8): wfold[0:72], wfold[72] = cfold
  const float* fold_u; const float* fold_w; int64_t fold_ldw; const float* fold_b; const float* fold_b2;
  float* wfold;
};

__global__ void __launch_bounds__(kThreads)
ncfp_prep_kernel(const Prep A) {
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int64_t wave = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6;
  const int64_t ublocks = (A.nu + 15) / 16, iblocks = (A.ni + 15) / 16;
  if (A.counts) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < A.ncounts; i += (int64_t)gridDim.x * kThreads)
      A.counts[i] = 0;
  }
  if (blockIdx.x == gridDim.x - 1) {
    // the folded head: thread t < 64 copies u[t]; 64..71 column t - 64 of W^T u[64:]; 72 the folded bias
    const int t = threadIdx.x;
    if (t < kP) {
      A.wfold[t] = A.fold_u[t];
    } else if (t < kHeadW) {
      const float* wc = A.fold_w + (t - kP);
      const float* u = A.fold_u + kP;
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int i0 = 0; i0 < 64; i0 += 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(wc[(int64_t)(i0 + e) * A.fold_ldw], u[i0 + e], acc[e]);
      }
      A.wfold[t] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    } else if (t == kHeadW) {
      const float* u = A.fold_u + kP;
      float acc = A.fold_b2 ? A.fold_b2[0] : 0.0f;
      if (A.fold_b)
        for (int i = 0; i < 64; ++i) acc = fmaf(A.fold_b[i], u[i], acc);
      A.wfold[kHeadW] = acc;
    }
  }
  if (wave >= ublocks + iblocks) return;
  // one wave = sixteen table rows: out^T (64 units x 16 rows) = W0half (64 x 64) . X^T (64 x 16 rows)
  const bool user = wave < ublocks;
  const int64_t row = (user ? wave : wave - ublocks) * 16 + n, rows = user ? A.nu : A.ni;
  const float* tab = user ? A.mlp_u : A.mlp_i;
  const int coff = user ? 0 : kH;
  const bool live = row < rows;
  f32x4 x[4], acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] = ldg4(tab + (live ? row : 0) * kH + 16 * j + 4 * q);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (!user && A.b0) acc[b] = ldg4(A.b0 + 16 * b + 4 * q);
    else acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 w[4];   // A operands straight from the weight rows: W0[16b + n][coff + 16j + 4q + c]
#pragma unroll
    for (int b = 0; b < 4; ++b) w[b] = ldg4(A.w0 + (int64_t)(16 * b + n) * A.ldw0 + coff + 16 * j + 4 * q);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[b][c], x[j][c], acc[b], 0, 0, 0);
  }
  if (live) {
    float* dst = A.ptab + ((user ? 0 : A.nu) + row) * kN0 + 4 * q;
#pragma unroll
    for (int b = 0; b < 4; ++b) stg4(dst + 16 * b, acc[b]);
  }
}

// ------------------------------------------------------------------ forward
struct Fwd {
  Ids ids;
  const float* ptab;                           // (nu + ni, 64)
  const float* gmf_u; const float* gmf_i;      // (nu, 64), (ni, 64)
  const float* wfold;                          // 72 weights + the bias
  float* out; int64_t ldout; int act;
  int32_t* err_flag;
  int32_t* counts;                             // nullable (inference): per-row sample counters, users then items
  int32_t* ranks;                              // (2, m): rank of a sample inside its user row / item row (-1: bad id)
};

__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 4)))
ncfp_fwd_kernel(const Tower T, int64_t m, const Fwd F) {
  __shared__ __attribute__((aligned(16))) float s_w[kWFloats];
  __shared__ __attribute__((aligned(16))) float s_b[kBFloats];
  __shared__ __attribute__((aligned(16))) float s_hw[kHeadW + 4];
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int64_t groups = (m + 15) / 16;
  const int64_t wave0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * kThreads) >> 6;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const float* pu = F.ptab;
  const float* pi = F.ptab + F.ids.nu * kN0;
  // operands of a group: ids -> (rank atomics) -> rows; the next group's are requested while this one computes
  int64_t un = 0, in_ = 0;
  auto fetch_ids = [&](int64_t g) {
    const int64_t row = g * 16 + n;
    un = in_ = 0;
    if (g < groups && row < m) {
      un = F.ids.uidx[row * F.ids.ustride];
      in_ = F.ids.iidx[row * F.ids.istride];
    }
  };
  struct Ops {
    f32x4 a[4];      // P_U + P_I, columns 16j + 4q ..
    f32x4 xe[4];     // GMF_U * GMF_I, columns 16q + 4i ..
    int rank;        // q == 0: rank in the user row, q == 1: in the item row
  };
  auto fetch = [&](int64_t g, Ops& o) {
    const int64_t row = g * 16 + n;
    const bool live = g < groups && row < m;
    int64_t u = un, i = in_;
    const bool ubad = u < 0 || u >= F.ids.nu, ibad = i < 0 || i >= F.ids.ni;
    if (live && (ubad || ibad) && F.err_flag) *F.err_flag = 1;
    if (ubad) u = 0;
    if (ibad) i = 0;
    o.rank = -1;
    if (F.counts && live) {
      if (q == 0 && !ubad) o.rank = atomicAdd(F.counts + u, 1);
      if (q == 1 && !ibad) o.rank = atomicAdd(F.counts + F.ids.nu + i, 1);
    }
    const float* ru = pu + u * kN0 + 4 * q;
    const float* ri = pi + i * kN0 + 4 * q;
    const float* gu = F.gmf_u + u * kP + 16 * q;
    const float* gi = F.gmf_i + i * kP + 16 * q;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.a[j] = live ? ldg4(ru + 16 * j) + ldg4(ri + 16 * j) : zero4;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.xe[j] = live ? ldg4(gu + 4 * j) * ldg4(gi + 4 * j) : zero4;
  };
  Ops cur, nxt;
  fetch_ids(wave0);
  {
    f32x4 wv[kFStagePer];
    int wdst[kFStagePer];
    float bv;
    stage_forward_load(T, wv, wdst, bv);
    float hw = 0.0f;
    if (threadIdx.x <= kHeadW) hw = F.wfold[threadIdx.x];
    fetch(wave0, cur);
    fetch_ids(wave0 + nwaves);
    stage_forward_store(s_w, s_b, wv, wdst, bv);
    if (threadIdx.x <= kHeadW) s_hw[threadIdx.x] = hw;
  }
  __syncthreads();
  const float hc = s_hw[kHeadW];
  for (int64_t g = wave0; g < groups; g += nwaves) {
    const int64_t row = g * 16 + n;
    const bool live = row < m;
    fetch(g + nwaves, nxt);
    fetch_ids(g + 2 * nwaves);
    if (F.ranks && live && q < 2) F.ranks[(int64_t)q * m + row] = cur.rank;
    f32x4 a0[4], y1[2], y2[1], y3[1];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) a0[j][r] = fmaxf(cur.a[j][r], 0.0f);
    layer_fwd<0, 4>(s_w, s_b, lane, q, a0, y1);
    if (live) {
#pragma unroll
      for (int b = 0; b < 2; ++b) stg4(T.y[0] + row * T.ldy[0] + 16 * b + 4 * q, y1[b]);
    }
    layer_fwd<1, 2>(s_w, s_b, lane, q, y1, y2);
    if (live) stg4(T.y[1] + row * T.ldy[1] + 4 * q, y2[0]);
    layer_fwd<2, 1>(s_w, s_b, lane, q, y2, y3);
    if (live && q < 2) stg4(T.y[2] + row * T.ldy[2] + 4 * q, y3[0]);
    // head: prob = act([gmf | h] . wfold + cfold); the four lanes of a sample hold 16 + (q < 2 ? 4 : 0) terms each
    float dot = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(s_hw + 16 * q + 4 * i);
      dot = fmaf(cur.xe[i][0], wv[0], dot); dot = fmaf(cur.xe[i][1], wv[1], dot);
      dot = fmaf(cur.xe[i][2], wv[2], dot); dot = fmaf(cur.xe[i][3], wv[3], dot);
    }
    if (q < 2) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(s_hw + kP + 4 * q);
      dot = fmaf(y3[0][0], wv[0], dot); dot = fmaf(y3[0][1], wv[1], dot);
      dot = fmaf(y3[0][2], wv[2], dot); dot = fmaf(y3[0][3], wv[3], dot);
    }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    if (q == 0 && live) F.out[row * F.ldout] = ctr_act(dot + hc, F.act);
    cur = nxt;
  }
}

// ------------------------------------------------------------------ backward, per sample
// slab a workgroup leaves in the workspace: [dW_l | db_l] for the three layers, then the head's sums in the layout
// ctr_reduce_segments_fold expects (64 zeros -- the GMF part comes from the table rows, ncfp_finish -- then
// sum gz * h (8) and sum gz)
constexpr int slab_w(int l) {
  int o = 0;
  for (int i = 0; i < l; ++i) o += kN[i] * kK[i] + kN[i];
  return o;
}
constexpr int kSlabHead = slab_w(kL);                     // 2744
constexpr int kSlab = kSlabHead + kHeadW + 1;             // 2817
constexpr int kVecs = 8 + 2 + 1;                          // dW accumulator vectors of a lane
constexpr int kSmall = 32 + 16 + 8 + 8 + 1;               // bias sums, sum gz * h, sum gz
constexpr int kCopy = kVecs * 256 + 68;                   // one wave's sums parked in LDS (16-byte multiple)
constexpr int kStripP = 3 * kTile;                        // per wave: tiles A0 A1 | B0

struct Bwd {
  Ids ids;
  const float* ptab;
  const float* wfold;
  const float* prob; int64_t ldp;
  const float* gprob; int64_t ldgp;
  int act;
  const int32_t* counts;                       // (nu + ni) from the forward
  const int32_t* ranks;                        // (2, m)
  float* gz;                                   // (2m, 64): buckets, user rows' slots first
  float* aux;                                  // (2m, 4): {gz, partner id, row, -} per slot
  int32_t* offsets;                            // (nu + ni + 1): written by workgroup 0 for the later launches
  float* slabs;                                // (grid, kSlab)
  float* zero_a; int64_t zero_a_floats;        // cleared first: the segment sums (nu + ni, 128)
  float* zero_b; int64_t zero_b_floats;        // cleared first (nullable): the step's gradient buffer
};

__global__ void __launch_bounds__(kThreads)
ncfp_bwd_kernel(const Tower T, int64_t m, const Bwd B) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ __attribute__((aligned(16))) float s_hw[kHeadW + 4];
  __shared__ int s_scan[kWaves];
  float* s_wt = lds;
  const int lane = threadIdx.x & 63, q = lane >> 4, lo = lane & 15, wave = threadIdx.x >> 6;
  float* tA = lds + kWFloats + wave * kStripP;
  float* tB = tA + 2 * kTile;
  const int64_t nrows = B.ids.nu + B.ids.ni;
  int* s_off = reinterpret_cast<int*>(lds + kWFloats + kWaves * kStripP);   // nrows + 1 exclusive offsets
  const int64_t groups = (m + 15) / 16;
  const int64_t wave0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * kThreads) >> 6;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const float* pu = B.ptab;
  const float* pi = B.ptab + B.ids.nu * kN0;

  // ---- clear what this call accumulates into (both 16-byte aligned multiples of 4 floats)
  {
    const int64_t t0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 4, step = (int64_t)gridDim.x * kThreads * 4;
    for (int64_t i = t0; i < B.zero_a_floats; i += step) stg4(B.zero_a + i, zero4);
    if (B.zero_b)
      for (int64_t i = t0; i < B.zero_b_floats; i += step) stg4(B.zero_b + i, zero4);
  }
  // ---- operands of a sample group (sample-major "d": this lane's sample lo, units 4q + r; unit-major "t": unit lo,
  // samples 4q + c), requested a group ahead
  struct Ops {
    float gp, pb;
    f32x4 y3d, y2d, y1d[2], a0d[4];
    f32x4 y2t, y1t[2], a0t[4];
    int su, si;          // bucket slots of this lane's sample (-1: none)
    int uu, ii;          // its ids (row / partner of the slots)
  };
  auto fetch = [&](int64_t g, Ops& o) {
    const int64_t row = g * 16 + lo;
    const bool live = g < groups && row < m;
    const int64_t rc = live ? row : m - 1;
    int64_t rt[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t r = g * 16 + 4 * q + c;
      rt[c] = (g >= groups || r >= m) ? m - 1 : r;      // (a clamped row meets a zero gradient)
    }
    o.gp = live ? B.gprob[rc * B.ldgp] : 0.0f;           // a dead lane's gz is zero: it adds nothing anywhere
    o.pb = B.prob[rc * B.ldp];
    int64_t u = B.ids.uidx[rc * B.ids.ustride], i = B.ids.iidx[rc * B.ids.istride];
    const bool ubad = u < 0 || u >= B.ids.nu, ibad = i < 0 || i >= B.ids.ni;
    if (ubad) u = 0;
    if (ibad) i = 0;
    o.uu = (int)u; o.ii = (int)i;
    const int ru = B.ranks[rc], ri = B.ranks[m + rc];
    o.su = (live && !ubad && ru >= 0) ? ru : -1;          // + offset once the scan is there
    o.si = (live && !ibad && ri >= 0) ? ri : -1;
    o.y3d = q < 2 ? ldg4(T.y[2] + rc * T.ldy[2] + 4 * q) : zero4;
    o.y2d = ldg4(T.y[1] + rc * T.ldy[1] + 4 * q);
#pragma unroll
    for (int b = 0; b < 2; ++b) o.y1d[b] = ldg4(T.y[0] + rc * T.ldy[0] + 16 * b + 4 * q);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 z = ldg4(pu + u * kN0 + 16 * j + 4 * q) + ldg4(pi + i * kN0 + 16 * j + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) o.a0d[j][r] = fmaxf(z[r], 0.0f);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      o.y2t[c] = T.y[1][rt[c] * T.ldy[1] + lo];
#pragma unroll
      for (int b = 0; b < 2; ++b) o.y1t[b][c] = T.y[0][rt[c] * T.ldy[0] + 16 * b + lo];
      int64_t uc = B.ids.uidx[rt[c] * B.ids.ustride], ic = B.ids.iidx[rt[c] * B.ids.istride];
      if (uc < 0 || uc >= B.ids.nu) uc = 0;
      if (ic < 0 || ic >= B.ids.ni) ic = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) o.a0t[j][c] = fmaxf(pu[uc * kN0 + 16 * j + lo] + pi[ic * kN0 + 16 * j + lo], 0.0f);
    }
  };
  Ops cur, nxt;
  // ---- what a lane sums over every group it walks
  f32x4 dw0[2][4], dw1[2], dw2;                // dW blocks: register r = row 4q + r, column lo
  f32x4 sb0[2], sb1, sb2, hy = zero4;          // bias sums of this lane's sample: units 4q + r
  float hc = 0.0f;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    sb0[b] = zero4;
    dw1[b] = zero4;
#pragma unroll
    for (int j = 0; j < 4; ++j) dw0[b][j] = zero4;
  }
  sb1 = sb2 = dw2 = zero4;
  {
    f32x4 wv[kStagePer];
    int wdst[kStagePer];
    stage_transposed_load(T, wv, wdst);
    fetch(wave0, cur);
    stage_transposed_store(s_wt, wv, wdst);
  }
  if (threadIdx.x <= kHeadW) s_hw[threadIdx.x] = B.wfold[threadIdx.x];
  // ---- exclusive scan of the per-row sample counts (users, then items) into s_off: every workgroup for itself
  {
    const int per = (int)((nrows + kThreads - 1) / kThreads);
    const int64_t i0 = (int64_t)threadIdx.x * per;
    int sum = 0;
    for (int e = 0; e < per; ++e)
      if (i0 + e < nrows) sum += B.counts[i0 + e];
    // inclusive scan of `sum` over the workgroup's threads
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(inc, d, 64);
      if (lane >= d) inc += v;
    }
    if (lane == 63) s_scan[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_scan[w];
    int run = base + inc - sum;
    for (int e = 0; e < per; ++e)
      if (i0 + e < nrows) {
        s_off[i0 + e] = run;
        run += B.counts[i0 + e];
      }
    if (threadIdx.x == kThreads - 1) s_off[nrows] = base + inc;
    __syncthreads();
    if (blockIdx.x == 0)
      for (int64_t i = threadIdx.x; i <= nrows; i += kThreads) B.offsets[i] = s_off[i];
  }

  for (int64_t g = wave0; g < groups; g += nwaves) {
    fetch(g + nwaves, nxt);
    // ---- head: gz, the head's sums, the tower's (masked) gY
    const float gzs = cur.gp * ctr_act_grad(cur.pb, B.act);
    if (q == 0) hc += gzs;
    f32x4 gz3[1];
    {
      const f32x4 wv = q < 2 ? *reinterpret_cast<const f32x4*>(s_hw + kP + 4 * q) : zero4;
      hy += gzs * cur.y3d;                                    // (y3d is zero for q >= 2)
      gz3[0] = relu_mask(gzs * wv, cur.y3d);
    }
    sb2 += gz3[0];
    tiles_put<1>(tA, q, lo, gz3);
    f32x4 w0 = zero4, w1 = zero4;
    dx_first<2>(s_wt, lane, w0, w1);
    // ---- layer 2 (16 -> 8)
    f32x4 gz2[1];
    {
      f32x4 tg[1];
      tiles_get<1>(tA, q, lo, tg);
      dx_layer<2, 1>(s_wt, lane, gz3, w0, w1, [&](int, const f32x4& d0, const f32x4&) { gz2[0] = relu_mask(d0, cur.y2d); });
      sb1 += gz2[0];
      tiles_put<1>(tB, q, lo, gz2);
      dx_first<1>(s_wt, lane, w0, w1);
#pragma unroll
      for (int c = 0; c < 4; ++c) dw2 = __builtin_amdgcn_mfma_f32_16x16x4f32(tg[0][c], cur.y2t[c], dw2, 0, 0, 0);
    }
    // ---- layer 1 (32 -> 16)
    f32x4 gz1[2];
    {
      f32x4 tg[1];
      tiles_get<1>(tB, q, lo, tg);
      dx_layer<1, 1>(s_wt, lane, gz2, w0, w1, [&](int, const f32x4& d0, const f32x4& d1) {
        gz1[0] = relu_mask(d0, cur.y1d[0]);
        gz1[1] = relu_mask(d1, cur.y1d[1]);
      });
#pragma unroll
      for (int b = 0; b < 2; ++b) sb0[b] += gz1[b];
      tiles_put<2>(tA, q, lo, gz1);
      dx_first<0>(s_wt, lane, w0, w1);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) dw1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(tg[0][c], cur.y1t[j][c], dw1[j], 0, 0, 0);
    }
    // ---- layer 0 (64 -> 32): its input gradient, masked by relu'(z0), IS gz0 -- stored into the sample's two buckets
    {
      f32x4 tg[2];
      tiles_get<2>(tA, q, lo, tg);
      const int su = cur.su < 0 ? -1 : cur.su + s_off[cur.uu];
      const int si = cur.si < 0 ? -1 : cur.si + s_off[B.ids.nu + cur.ii];
      float* du = B.gz + (int64_t)su * kN0 + 4 * q;
      float* di = B.gz + (int64_t)si * kN0 + 4 * q;
      dx_layer<0, 2>(s_wt, lane, gz1, w0, w1, [&](int j, const f32x4& d0, const f32x4& d1) {
        const f32x4 g0 = relu_mask(d0, cur.a0d[j]), g1 = relu_mask(d1, cur.a0d[j + 1]);
        if (su >= 0) {
          stg4(du + 16 * j, g0);
          stg4(du + 16 * (j + 1), g1);
        }
        if (si >= 0) {
          stg4(di + 16 * j, g0);
          stg4(di + 16 * (j + 1), g1);
        }
      });
      if (q == 0) {
        if (su >= 0) stg4(B.aux + (int64_t)su * 4, f32x4{gzs, __int_as_float(cur.ii), __int_as_float(cur.uu), 0.0f});
        if (si >= 0)
          stg4(B.aux + (int64_t)si * 4, f32x4{gzs, __int_as_float(cur.uu), __int_as_float((int)B.ids.nu + cur.ii), 0.0f});
      }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            dw0[b][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(tg[b][c], cur.a0t[j][c], dw0[b][j], 0, 0, 0);
    }
    cur = nxt;
  }

  // ---- the workgroup's partial: every wave parks its sums (the weights and tiles are dead), then the slab is summed
  // over the four copies on the way out
  __syncthreads();
  {
    auto rsum = [&](const f32x4& v) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = row_sum16(v[r]);
      return o;
    };
    float* copy = lds + wave * kCopy;
    float* small = copy + kVecs * 256;
    auto vec = [&](int v) -> const f32x4& { return v < 8 ? dw0[v >> 2][v & 3] : v < 10 ? dw1[v - 8] : dw2; };
#pragma unroll
    for (int v = 0; v < kVecs; ++v) *reinterpret_cast<f32x4*>(copy + (v * 64 + lane) * 4) = vec(v);
    const f32x4 s00 = rsum(sb0[0]), s01 = rsum(sb0[1]), s1 = rsum(sb1), s2 = rsum(sb2), shy = rsum(hy);
    const float vc = row_sum16(hc);
    if (lo == 0) {
      *reinterpret_cast<f32x4*>(small + 4 * q) = s00;              // layer 0 bias sums: units 4q .. (block 0)
      *reinterpret_cast<f32x4*>(small + 16 + 4 * q) = s01;         //                     16 + 4q ..
      *reinterpret_cast<f32x4*>(small + 32 + 4 * q) = s1;          // layer 1: 16 units
      if (q < 2) {
        *reinterpret_cast<f32x4*>(small + 48 + 4 * q) = s2;        // layer 2: 8 units
        *reinterpret_cast<f32x4*>(small + 56 + 4 * q) = shy;       // sum gz * h: 8
      }
    }
    if (lane == 0) small[64] = vc;
  }
  __syncthreads();
  float* out = B.slabs + (int64_t)blockIdx.x * kSlab;
  auto sum4 = [&](int at) { return (lds[at] + lds[kCopy + at]) + (lds[2 * kCopy + at] + lds[3 * kCopy + at]); };
  // dW vectors: vector v of lane (q, lo), register r  ->  row 16b + 4q + r, column 16j + lo of its layer
  for (int e = threadIdx.x; e < kVecs * 256; e += kThreads) {
    const int v = e >> 8, ln = (e >> 2) & 63, r = e & 3, qq = ln >> 4, ll = ln & 15;
    const int l = v < 8 ? 0 : v < 10 ? 1 : 2;
    const int vv = v - (l == 0 ? 0 : l == 1 ? 8 : 10);
    const int J = l == 0 ? 4 : l == 1 ? 2 : 1, K = l == 0 ? kK[0] : l == 1 ? kK[1] : kK[2];
    const int bb = vv / J, jj = vv - bb * J;
    const int rowi = 16 * bb + 4 * qq + r;
    if (rowi < (l == 0 ? kN[0] : l == 1 ? kN[1] : kN[2]))
      out[(l == 0 ? slab_w(0) : l == 1 ? slab_w(1) : slab_w(2)) + rowi * K + 16 * jj + ll] = sum4(e);
  }
  for (int i = threadIdx.x; i < kSmall; i += kThreads) {
    const int at = kVecs * 256 + i;
    int dst;
    if (i < 32) dst = slab_w(0) + kN[0] * kK[0] + i;
    else if (i < 48) dst = slab_w(1) + kN[1] * kK[1] + (i - 32);
    else if (i < 56) dst = slab_w(2) + kN[2] * kK[2] + (i - 48);
    else dst = kSlabHead + kP + (i - 56);                          // 8 x (gz * h), then sum gz
    out[dst] = sum4(at);
  }
  for (int i = threadIdx.x; i < kP; i += kThreads) out[kSlabHead + i] = 0.0f;   // the GMF part: ncfp_finish
}

// ------------------------------------------------------------------ segment sums over the buckets
// ST[v] = [ S[v] (64) | T[v] (64) ],  S[v] = sum of the gz0 rows in row v's bucket,  T[v] = sum gz_b * partner row.
// A lane group of sixteen owns sixteen consecutive SLOTS, whatever rows they belong to: equal work per wave under any
// id distribution.  It keeps running sums for the row it is in and adds them to ST when the row changes (and at its
// end): one 64-byte atomic segment per sixteen lanes and quarter row, after a transposition through LDS (lane lo of
// the loads holds columns 4lo .. 4lo+3; an atomic instruction wants sixteen consecutive floats from sixteen lanes).
struct Seg {
  const float* gz; const float* aux; const int32_t* offsets;
  const float* gmf_u; const float* gmf_i;
  int64_t nu, ni;
  float* st;                                   // (nu + ni, 128), zeroed
};

__global__ void __launch_bounds__(kThreads)
ncfp_segsum_kernel(const Seg A) {
  __shared__ __attribute__((aligned(16))) float s_t[kThreads / 16][2][68];
  const int lane = threadIdx.x & 63, lo = lane & 15, grp = threadIdx.x >> 4;
  const int64_t total = A.offsets[A.nu + A.ni];
  const int64_t s0 = ((int64_t)blockIdx.x * (kThreads / 16) + grp) * 16;
  if (s0 >= total) return;
  const int cnt = (int)((total - s0) < 16 ? (total - s0) : 16);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 accs = zero4, acct = zero4;
  int cur = -1;
  auto flush = [&](int v) {
    float* t = &s_t[grp][0][0];
    *reinterpret_cast<f32x4*>(t + 4 * lo) = accs;
    *reinterpret_cast<f32x4*>(t + 68 + 4 * lo) = acct;
    // (same lane group, same wave: the LDS pipe is in order, no barrier needed; the clobbers pin the order)
    asm volatile("" ::: "memory");
    float* dst = A.st + (int64_t)v * 128;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ctr_atomic_add_global(dst + 16 * r + lo, t[16 * r + lo]);
      ctr_atomic_add_global(dst + 64 + 16 * r + lo, t[68 + 16 * r + lo]);
    }
    asm volatile("" ::: "memory");
  };
#pragma unroll
  for (int k0 = 0; k0 < 16; k0 += 4) {
    f32x4 ax[4], g[4], p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t s = s0 + (k0 + k < cnt ? k0 + k : cnt - 1);
      ax[k] = ldg4(A.aux + s * 4);
      g[k] = ldg4(A.gz + s * kN0 + 4 * lo);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pid = __float_as_int(ax[k][1]), v = __float_as_int(ax[k][2]);
      const float* prow = (v < A.nu ? A.gmf_i : A.gmf_u) + (int64_t)pid * kP;
      p[k] = ldg4(prow + 4 * lo);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k0 + k < cnt) {
        const int v = __float_as_int(ax[k][2]);
        if (v != cur) {
          if (cur >= 0) flush(cur);
          cur = v;
          accs = acct = zero4;
        }
        accs += g[k];
        acct += ax[k][0] * p[k];
      }
    }
  }
  if (cur >= 0) flush(cur);
}

// ------------------------------------------------------------------ the products over the table rows
// one wave = sixteen rows of one table:  dMLP[rows] += S . W0half,  dGMF[rows] += wf * T,  and its share of
// dW0half += S^T . MLP,  db0 += column sums of S (user rows),  g_head_w[:64] += sum_rows GMF * T (user rows); the four
// waves of a workgroup (same table) meet in LDS, one atomic per element and workgroup goes out.
struct Fin {
  const float* st;                             // (nu + ni, 128)
  const float* mlp_u; const float* mlp_i; const float* gmf_u; const float* gmf_i;
  const float* w0; int64_t ldw0;
  const float* wfold;
  int64_t nu, ni;
  float* g_mlp_u; float* g_mlp_i; float* g_gmf_u; float* g_gmf_i;   // (+=), nullable
  float* g_w0; int64_t ldgw0; float* g_b0;                           // (+=), nullable
  float* g_head;                                                     // g of linear2.weight[:64] (+=), nullable
};

__global__ void __launch_bounds__(kThreads)
ncfp_finish_kernel(const Fin A) {
  __shared__ __attribute__((aligned(16))) float s_dw[2][16 * 256];
  __shared__ float s_sm[kWaves][2][64];
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15, wave = threadIdx.x >> 6;
  const int64_t ublocks = (A.nu + 15) / 16, iblocks = (A.ni + 15) / 16;
  const int64_t uwgs = (ublocks + kWaves - 1) / kWaves;
  const bool user = (int64_t)blockIdx.x < uwgs;
  const int64_t blk = (user ? (int64_t)blockIdx.x : (int64_t)blockIdx.x - uwgs) * kWaves + wave;
  const int64_t rows = user ? A.nu : A.ni, r0 = blk * 16;
  const bool any = blk < (user ? ublocks : iblocks);
  const float* st = A.st + (user ? 0 : A.nu) * 128;
  const float* tab = user ? A.mlp_u : A.mlp_i;
  const float* gmf = user ? A.gmf_u : A.gmf_i;
  float* gtab = user ? A.g_mlp_u : A.g_mlp_i;
  float* ggmf = user ? A.g_gmf_u : A.g_gmf_i;
  const int coff = user ? 0 : kH;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t row = r0 + n;
  const bool live = any && row < rows;
  f32x4 dw[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) dw[b][j] = zero4;
  f32x4 colsum[4], gw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) colsum[j] = gw[j] = zero4;
  if (any) {
    // ---- sample-major operands: this lane's row n, columns 16j + 4q ..
    f32x4 sd[4], td[4], gd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sd[j] = live ? ldg4(st + row * 128 + 16 * j + 4 * q) : zero4;
      td[j] = live ? ldg4(st + row * 128 + 64 + 16 * j + 4 * q) : zero4;
      gd[j] = live ? ldg4(gmf + row * kP + 16 * j + 4 * q) : zero4;
    }
    // dGMF[row] += wfold[:64] * T[row];   sum_rows GMF * T
    if (ggmf && live) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* d = ggmf + row * kP + 16 * j + 4 * q;
        stg4(d, ldg4(d) + ldg4(A.wfold + 16 * j + 4 * q) * td[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gw[j] = gd[j] * td[j];
      colsum[j] = sd[j];
    }
    // ---- dMLP^T (64 inputs x 16 rows) = W0half^T (64 x 64 units) . S^T (64 units x 16 rows)
    if (gtab) {
      f32x4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 wt[4];   // A operands: W0[16j + 4q + c][coff + 16b + n]
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int c = 0; c < 4; ++c) wt[b][c] = A.w0[(int64_t)(16 * j + 4 * q + c) * A.ldw0 + coff + 16 * b + n];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[b][c], sd[j][c], acc[b], 0, 0, 0);
      }
      if (live) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          float* d = gtab + row * kH + 16 * b + 4 * q;
          stg4(d, ldg4(d) + acc[b]);
        }
      }
    }
    // ---- dW0half (64 units x 64 inputs) += S^T . X: unit-major operands, unit / input n of rows 4q + c
    if (A.g_w0) {
      f32x4 stt[4], xt[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int64_t r = r0 + 4 * q + c;
        const bool ok = r < rows;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          stt[b][c] = ok ? st[r * 128 + 16 * b + n] : 0.0f;
          xt[b][c] = ok ? tab[r * kH + 16 * b + n] : 0.0f;
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) dw[b][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(stt[b][c], xt[j][c], dw[b][j], 0, 0, 0);
    }
  }
  // ---- the workgroup's sums: waves 0 / 1 park their dW blocks, waves 2 / 3 add theirs in place (own lane slots)
  if (wave < 2) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&s_dw[wave][((b * 4 + j) * 64 + lane) * 4]) = dw[b][j];
  }
  __syncthreads();
  if (wave >= 2) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4* at = reinterpret_cast<f32x4*>(&s_dw[wave - 2][((b * 4 + j) * 64 + lane) * 4]);
        *at = *at + dw[b][j];
      }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float cs = row_sum16(colsum[j][r]), gs = row_sum16(gw[j][r]);
      if (n == 0) {
        s_sm[wave][0][16 * j + 4 * q + r] = cs;
        s_sm[wave][1][16 * j + 4 * q + r] = gs;
      }
    }
  __syncthreads();
  if (A.g_w0) {
    for (int e = threadIdx.x; e < 64 * 64; e += kThreads) {
      const int unit = e >> 6, k = e & 63;                     // dW0[unit][coff + k]
      const int b = unit >> 4, qq = (unit >> 2) & 3, r = unit & 3, j = k >> 4, ll = k & 15;
      const int at = ((b * 4 + j) * 64 + qq * 16 + ll) * 4 + r;
      const float v = s_dw[0][at] + s_dw[1][at];
      ctr_atomic_add_global(A.g_w0 + (int64_t)unit * A.ldgw0 + coff + k, v);
    }
  }
  if (threadIdx.x < 64 && user) {
    const int t = threadIdx.x;
    if (A.g_b0) ctr_atomic_add_global(A.g_b0 + t, (s_sm[0][0][t] + s_sm[1][0][t]) + (s_sm[2][0][t] + s_sm[3][0][t]));
    if (A.g_head) ctr_atomic_add_global(A.g_head + t, (s_sm[0][1][t] + s_sm[1][1][t]) + (s_sm[2][1][t] + s_sm[3][1][t]));
  }
}

int fill_tower(Tower* T, const ctr_mlp_layer_t* layers, bool need_y) {
  for (int l = 0; l < kL; ++l) {
    const ctr_mlp_layer_t& s = layers[l + 1];
    if (s.n != kN[l] || s.k != kK[l] || s.act != CTR_ACT_RELU || !s.w || !ctr_aligned16(s.w)) return CTR_ELIMIT;
    if (need_y && (!s.y || !ctr_aligned16(s.y) || s.ldy % 4 != 0 || s.ldy < s.n)) return CTR_ELIMIT;
    T->w[l] = s.w; T->b[l] = s.b; T->y[l] = s.y; T->ldy[l] = s.ldy;
  }
  return CTR_OK;
}

bool pattern_ok(const ctr_ncf_proj_t* d) {
  const ctr_mlp_layer_t& l0 = d->layers[0];
  return d->user_idx && d->item_idx && d->mlp_user && d->mlp_item && d->gmf_user && d->gmf_item && l0.w && l0.n == kN0 &&
         l0.k == 2 * kH && l0.act == CTR_ACT_RELU && ctr_aligned16(l0.w) && d->mlp_dim == kH && d->mf_dim == kP &&
         d->proj_n == kP && d->proj_k == kNL && d->proj_w && d->head_w && d->num_users >= 1 && d->num_items >= 1 &&
         d->num_users + d->num_items <= CTR_NCF_PROJ_MAX_ROWS && ctr_aligned16(d->mlp_user) && ctr_aligned16(d->mlp_item) &&
         ctr_aligned16(d->gmf_user) && ctr_aligned16(d->gmf_item) && d->ptab && ctr_aligned16(d->ptab) && d->wfold &&
         ctr_aligned16(d->wfold) && d->batch * 2 < ((int64_t)1 << 31);
}

}  // namespace

static int64_t workspace_floats(int64_t batch, int64_t num_users, int64_t num_items) {
  const int64_t rows = num_users + num_items;
  const int64_t groups = ctr_ceil_div(batch > 0 ? batch : 1, 16);
  int64_t grid = ctr_ceil_div(groups, kWaves);
  if (grid > 768) grid = 768;
  // buckets (2B, 64) | slot records (2B, 4) | segment sums (rows, 128) | 128 scratch | offsets (rows + 1) | slabs
  return 2 * batch * kN0 + 2 * batch * 4 + rows * 128 + 128 + (rows + 1 + 3) / 4 * 4 + grid * (int64_t)kSlab;
}

extern "C" int ctr_ncf_proj_workspace_floats(int64_t batch, int64_t num_users, int64_t num_items, int64_t* floats) {
  CTR_REQUIRE(floats && batch >= 0 && num_users >= 0 && num_items >= 0, CTR_EINVAL);
  *floats = workspace_floats(batch, num_users, num_items);
  return CTR_OK;
}

extern "C" int ctr_ncf_proj_fwd(const ctr_ncf_proj_t* d, void* stream) {
  CTR_REQUIRE(d && d->batch >= 0, CTR_EINVAL);
  if (!pattern_ok(d)) return CTR_ELIMIT;
  CTR_REQUIRE(d->prob && d->ldprob >= 1 && d->head_act >= CTR_ACT_NONE && d->head_act <= CTR_ACT_SIGMOID, CTR_EINVAL);
  CTR_REQUIRE(d->ld_proj_w >= d->proj_k, CTR_EINVAL);
  CTR_REQUIRE(!d->training || (d->counts && d->ranks), CTR_EINVAL);
  Tower T;
  int rc = fill_tower(&T, d->layers, true);
  if (rc != CTR_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nu = d->num_users, ni = d->num_items;
  const int64_t pwaves = ctr_ceil_div(nu, 16) + ctr_ceil_div(ni, 16);
  const Prep P{d->mlp_user, d->mlp_item, d->layers[0].w, d->layers[0].k, d->layers[0].b, d->ptab, nu, ni,
               d->training ? d->counts : nullptr, nu + ni, d->head_w, d->proj_w, d->ld_proj_w, d->proj_b, d->head_b, d->wfold};
  hipLaunchKernelGGL(ncfp_prep_kernel, dim3((unsigned)(ctr_ceil_div(pwaves, kWaves) + 1)), dim3(kThreads), 0, st, P);
  rc = ctr_launch_status();
  if (rc != CTR_OK || d->batch == 0) return rc;
  const Fwd F{Ids{d->user_idx, d->user_stride, d->item_idx, d->item_stride, nu, ni}, d->ptab, d->gmf_user, d->gmf_item,
              d->wfold, d->prob, d->ldprob, d->head_act, d->err_flag, d->training ? d->counts : nullptr,
              d->training ? d->ranks : nullptr};
  const int64_t groups = ctr_ceil_div(d->batch, 16);
  int64_t grid = ctr_ceil_div(groups, kWaves);
  if (grid > 256 * 4) grid = 256 * 4;
  hipLaunchKernelGGL(ncfp_fwd_kernel, dim3((unsigned)grid), dim3(kThreads), 0, st, T, d->batch, F);
  return ctr_launch_status();
}

extern "C" int ctr_ncf_proj_bwd(const ctr_ncf_proj_t* d, const ctr_ncf_proj_grad_t* g, void* stream) {
  CTR_REQUIRE(d && g && d->batch >= 0, CTR_EINVAL);
  if (!pattern_ok(d)) return CTR_ELIMIT;
  CTR_REQUIRE(d->counts && d->ranks && d->prob && g->gprob && g->ldgprob >= 1 && g->workspace, CTR_EINVAL);
  CTR_REQUIRE(!g->zero_buf || (ctr_aligned16(g->zero_buf) && g->zero_floats % 4 == 0 && g->zero_floats >= 0), CTR_EALIGN);
  Tower T;
  int rc = fill_tower(&T, d->layers, true);
  if (rc != CTR_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t m = d->batch, nu = d->num_users, ni = d->num_items, rows = nu + ni;
  if (m == 0) return g->zero_buf ? ctr_zero_fill(g->zero_buf, g->zero_floats, st) : CTR_OK;
  for (int l = 1; l <= kL; ++l) CTR_REQUIRE(g->layers[l].gw && g->layers[l].gb, CTR_EINVAL);
  CTR_REQUIRE(g->workspace_floats >= workspace_floats(m, nu, ni) && ctr_aligned16(g->workspace), CTR_ELIMIT);
  // carve the workspace
  float* ws = g->workspace;
  float* gzb = ws;            ws += 2 * m * kN0;
  float* aux = ws;            ws += 2 * m * 4;
  float* stt = ws;            ws += rows * 128;
  float* scratch = ws;        ws += 128;   // cleared with the segment sums: the fold kernel accumulates its 73 sums here
  int32_t* offs = reinterpret_cast<int32_t*>(ws); ws += (rows + 1 + 3) / 4 * 4;
  float* slabs = ws;
  const int64_t groups = ctr_ceil_div(m, 16);
  int64_t grid = ctr_ceil_div(groups, kWaves);
  if (grid > 768) grid = 768;
  const Ids ids{d->user_idx, d->user_stride, d->item_idx, d->item_stride, nu, ni};
  const Bwd B{ids, d->ptab, d->wfold, d->prob, d->ldprob, g->gprob, g->ldgprob, d->head_act, d->counts, d->ranks, gzb, aux,
              offs, slabs, stt, rows * 128 + 128, g->zero_buf, g->zero_buf ? g->zero_floats : 0};
  const int64_t main_f = kWFloats + kWaves * kStripP + (rows + 1 + 3) / 4 * 4, copy_f = (int64_t)kWaves * kCopy;
  const size_t lds_bytes = sizeof(float) * (size_t)(main_f > copy_f ? main_f : copy_f);
  CTR_REQUIRE(lds_bytes <= 150 * 1024, CTR_ELIMIT);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(ncfp_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds_bytes) != hipSuccess)
    return CTR_ELAUNCH;
  hipLaunchKernelGGL(ncfp_bwd_kernel, dim3((unsigned)grid), dim3(kThreads), lds_bytes, st, T, m, B);
  rc = ctr_launch_status();
  if (rc != CTR_OK) return rc;
  // tower dW / db partials and the head fold's chain rule (its GMF part arrives from ncfp_finish)
  CtrSegments segs;
  segs.n = 0;
  int64_t off = 0;
  for (int l = 1; l <= kL; ++l) {
    const int64_t wn = (int64_t)d->layers[l].n * d->layers[l].k;
    segs.s[segs.n++] = CtrSegment{off, wn, g->layers[l].gw};
    segs.s[segs.n++] = CtrSegment{off + wn, d->layers[l].n, g->layers[l].gb};
    off += wn + d->layers[l].n;
  }
  const CtrHeadFoldGrad F{d->head_w, d->proj_w, d->ld_proj_w, d->proj_b, scratch, scratch + kHeadW, g->g_head_w, g->g_proj_w,
                          g->ld_g_proj_w, g->g_proj_b, g->g_head_b};
  rc = ctr_reduce_segments_fold(slabs, (int)grid, kSlab, segs, off, F, st);
  if (rc != CTR_OK) return rc;
  const Seg S{gzb, aux, offs, d->gmf_user, d->gmf_item, nu, ni, stt};
  hipLaunchKernelGGL(ncfp_segsum_kernel, dim3((unsigned)ctr_ceil_div(2 * m, kThreads)), dim3(kThreads), 0, st, S);
  rc = ctr_launch_status();
  if (rc != CTR_OK) return rc;
  const Fin N{stt, d->mlp_user, d->mlp_item, d->gmf_user, d->gmf_item, d->layers[0].w, d->layers[0].k, d->wfold, nu, ni,
              g->g_mlp_user, g->g_mlp_item, g->g_gmf_user, g->g_gmf_item, g->layers[0].gw, d->layers[0].k, g->layers[0].gb,
              g->g_head_w};
  const int64_t fwgs = ctr_ceil_div(ctr_ceil_div(nu, 16), kWaves) + ctr_ceil_div(ctr_ceil_div(ni, 16), kWaves);
  hipLaunchKernelGGL(ncfp_finish_kernel, dim3((unsigned)fwgs), dim3(kThreads), 0, st, N);
  return ctr_launch_status();
}
