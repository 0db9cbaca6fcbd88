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

#include <stdlib.h>

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

#ifdef CTR_STAMPS
// dev/ncfp_stamps.py: cycle stamps of wave 0 of workgroup 7 (a build of its own: the product has no stamp instruction)
__device__ unsigned long long g_stamps[2][64];
#define STAMP(k, i) do { const int at_ = (i); if (blockIdx.x == 7 && threadIdx.x == 0 && at_ < 64) g_stamps[k][at_] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(k, i) do {} while (0)
#endif

constexpr int kCountStride = CTR_NCF_PROJ_COUNT_STRIDE;   // int32 between two rows' sample counters

struct Ids {
  const int64_t* uidx; int64_t ustride;
  const int64_t* iidx; int64_t istride;
  int64_t nu, ni;
};

// Ranks of the samples [lo, hi): a returning atomic per sample and id column on the row's counter, an 8-byte record out.
// Run by workgroups `first` .. gridDim - 1 of a launch that has other work in its first workgroups (see struct Fwd).
struct RankJob {
  Ids ids;
  int32_t* counts; int32_t* ranks;
  int64_t lo, hi;
  int first;
};
__device__ __forceinline__ void rank_role(const RankJob& R) {
  const int64_t nb = (int64_t)gridDim.x - R.first;
  for (int64_t s0 = R.lo + ((int64_t)blockIdx.x - R.first) * kThreads + threadIdx.x; s0 < R.hi; s0 += nb * kThreads) {
    const int64_t u = R.ids.uidx[s0 * R.ids.ustride], i = R.ids.iidx[s0 * R.ids.istride];
    int ru = -1, ri = -1;                      // an id outside its table has no slot (the per-sample part raises the flag)
    if ((uint64_t)u < (uint64_t)R.ids.nu) ru = atomicAdd(R.counts + u * kCountStride, 1);
    if ((uint64_t)i < (uint64_t)R.ids.ni) ri = atomicAdd(R.counts + (R.ids.nu + i) * kCountStride, 1);
    *reinterpret_cast<int2*>(R.ranks + 2 * s0) = make_int2(ru, ri);
  }
}

// ------------------------------------------------------------------ prep: projected tables + head fold
struct Prep {
  const float* mlp_u; const float* mlp_i;      // (nu, 64), (ni, 64)
  const float* w0; int64_t ldw0; const float* b0;   // layer 0: (64, 128), (64)
  float* ptab;                                 // (nu + ni, 64): P_U rows, then P_I rows
  int64_t nu, ni;
  // head fold (ctr_fold_head_fwd's map for p = 64, n = 64, k = 8): wfold[0:72], wfold[72] = cfold
  const float* fold_u; const float* fold_w; int64_t fold_ldw; const float* fold_b; const float* fold_b2;
  float* wfold;
  RankJob rank;                                // training: the first part of the batch's ranks, by workgroups rank.first ..
};

__global__ void __launch_bounds__(kThreads)
ncfp_prep_kernel(const Prep A) {
  if ((int)blockIdx.x >= A.rank.first) {
    rank_role(A.rank);
    return;
  }
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int64_t wave = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6;
  const int64_t ublocks = (A.nu + 15) / 16, iblocks = (A.ni + 15) / 16;
  if ((int)blockIdx.x == A.rank.first - 1) {
    // the folded head (ctr_fold_head_fwd's map): wfold[t < 64] = u[t]; wfold[64 + c] = sum_i W[i][c] u[64 + i] (column c
    // by the 32 threads t % 8 == c, two terms each, summed through LDS); wfold[72] = b . u[64:] + b2
    __shared__ float s_f[kThreads];
    const int t = threadIdx.x;
    const float* u = A.fold_u + kP;
    if (t < kP) A.wfold[t] = A.fold_u[t];
    {
      const int c = t & 7, i = t >> 3;                 // i in 0..31: terms i and i + 32
      s_f[t] = fmaf(A.fold_w[(int64_t)i * A.fold_ldw + c], u[i], A.fold_w[(int64_t)(i + 32) * A.fold_ldw + c] * u[i + 32]);
    }
    __syncthreads();
    if (t < kNL) {
      float acc = 0.0f;
      for (int i = 0; i < 32; ++i) acc += s_f[8 * i + t];
      A.wfold[kP + t] = acc;
    }
    __syncthreads();
    if (t < 64) s_f[t] = A.fold_b ? A.fold_b[t] * u[t] : 0.0f;
    __syncthreads();
    if (t == 0) {
      float acc = A.fold_b2 ? A.fold_b2[0] : 0.0f;
      for (int i = 0; i < 64; ++i) acc += s_f[i];
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
  // training (counts != nullptr): workgroups fwd_blocks .. gridDim - 1 of the SAME launch take every sample's rank
  // inside its user row and its item row -- a returning atomic on counts (ALL ZERO at entry; the backward's last launch
  // leaves them zero again) -- and write ranks (m, 2) for the backward's bucketing.  They share the CUs with the
  // per-sample workgroups (two workgroups' worth of LDS fit a CU) and are done before those are.  Inside the per-sample
  // loop the same atomics cost 12 of the forward's 25.8 us: same-LINE atomics are served one after the other at the
  // memory side (~30 ns each) and a wave's memory operations retire in order (profiles/r03_rank_atomics.txt).  The unit
  // of that serialisation is the line, not the address: with the 2625 counters packed (164 lines, 800 adds each) the
  // atomics alone took 23 us; every counter therefore has a line to itself (kCountStride int32 apart): 6 us.
  // The batch's ranks are split over TWO launches: the first part beside the ~40 workgroups of the projection launch, the
  // rest here -- each about as long as the work it runs beside.
  RankJob rank;
};

// How the loops of this file are written (what the first version got wrong, found with cycle stamps, an ablation and
// the TA / TCP counters: dev/r03_ablate.sh, dev/ncfp_stamps.py, profiles/r03_ncfp_*):
//  * hipcc puts a `s_waitcnt vmcnt(0)` INSIDE every conditional block that consumes a load (`live ? a[i] + b[i] : 0`
//    becomes a branch around two loads, their wait and the add): a fetch written that way is a chain of round trips.
//    Every load here is unconditional (rows past the batch are clamped), every store too (lanes without a sample write
//    to a spare row every per-sample buffer has), so the compiler can count them.
//  * loading a gathered row straight into matrix-core operand layout -- lane (q, n) takes 16 bytes at column 16j + 4q
//    of sample n's row -- makes NEIGHBOURING lanes read DIFFERENT rows: 64 separate 16-byte requests per instruction
//    (TCP_TOTAL_CACHE_ACCESSES = 57 per vector-memory instruction, the texture addresser busy 75 % of the kernel).  The
//    rows are therefore fetched COALESCED -- sixteen lanes per 256-byte row, four rows per instruction -- by LDS-DMA
//    (global_load_lds_dwordx4: per-lane source address, wave-contiguous LDS image) into a wave-private stage, and read
//    back in operand layout with ds_read_b128; the source chunk a lane fetches is XOR-swizzled with its row so that the
//    sixteen rows a quarter-wave reads at one column land on sixteen different bank groups.
//  * the two dependent stages of a gather -- ids, then the rows they name -- are split over iterations: group g waits
//    once, reads its staged rows into registers, requests the rows of the next group (their ids arrived with that wait)
//    and the ids of the one after, and only then computes.
//
// LDS-DMA and its waits are written by hand (the compiler cannot see which bytes an LDS-DMA writes and would wait for
// every load in front of every LDS read).  Vector-memory operations retire in issue order; per group a wave issues
//   [kDma row fetches] [2 id loads] [kStores stores]
// so `vmcnt(kStores)` at the head of the next group means: its rows are staged and its successor's ids are here, while
// this group's stores may still be on their way.
__device__ __forceinline__ void dma16(const float* g, uint32_t lds_base) {
  // (m0 is a reserved register: the compiler only sets it right in front of an instruction that reads it)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_base) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const float* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}
constexpr int kFwdStage = 4 * 16 * 64;   // floats per wave: P_U, P_I, GMF_U, GMF_I rows of sixteen samples
constexpr int kFwdStores = 5;            // y1 x 2, y2, y3, prob

// DBG != 0: timing experiments (dev/r03_ablate.sh; results are wrong): 2 no layers, 4 no y stores, 16 no prob store, 32 no row fetch
template <int DBG>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
ncfp_fwd_kernel(const Tower T, int64_t m, const Fwd F) {
  if ((int)blockIdx.x >= F.rank.first) {
    rank_role(F.rank);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* s_w = lds;                                   // kWFloats
  float* s_b = s_w + kWFloats;                        // kBFloats
  float* s_hw = s_b + kBFloats;                       // kHeadW + 4 (+ 4 pad)
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* stage = s_hw + 80 + wave * kFwdStage;        // this wave's rows: [table][row][chunk ^ row] x 16 bytes
  const uint32_t stage_addr = lds_addr(stage);
  const int64_t groups = (m + 15) / 16;
  const int64_t wave0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6, nwaves = ((int64_t)F.rank.first * kThreads) >> 6;
  const uint32_t nu = (uint32_t)F.ids.nu, ni = (uint32_t)F.ids.ni;
  const float* tabs[4] = {F.ptab, F.ptab + F.ids.nu * kN0, F.gmf_u, F.gmf_i};
  // stage 1: the two ids of this lane's sample (sample n of the group, the same in its four lanes)
  int64_t idu = 0, idi = 0;
  auto issue_ids = [&](int64_t g) {
    int64_t row = g * 16 + n;
    row = row < m ? row : m - 1;
    if constexpr (DBG & 8) {
      idu = (row * 7) % F.ids.nu;
      idi = (row * 13) % F.ids.ni;
    } else {
      idu = F.ids.uidx[row * F.ids.ustride];
      idi = F.ids.iidx[row * F.ids.istride];
    }
  };
  // stage 2: the four rows of every sample of the group by LDS-DMA.  Instruction k of a table moves
  // rows 4k .. 4k+3: lane l fetches chunk (l % 16) ^ row of row 4k + l / 16 into slot (row, l % 16).
  auto issue_rows = [&](int64_t g) {
    const bool live = g * 16 + n < m;
    const bool ubad = (uint64_t)idu >= nu, ibad = (uint64_t)idi >= ni;
    const uint32_t u = ubad ? 0u : (uint32_t)idu, i = ibad ? 0u : (uint32_t)idi;
    if (live && (ubad || ibad) && F.err_flag) *F.err_flag = 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = 4 * k + q;                          // the row this lane helps to fetch
      const int src = (lane & 48) | r;                  // a lane of this quarter that holds sample r's ids
      const uint32_t ur = (uint32_t)__shfl((int)u, src, 64), ir = (uint32_t)__shfl((int)i, src, 64);
      const uint32_t col = 4u * (uint32_t)(n ^ r);      // swizzled 16-byte chunk of the row
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float* g_ = tabs[t] + ((t & 1) ? ir : ur) * 64u + col;
        if constexpr (DBG & 32) asm volatile("" ::"v"(g_));
        else dma16(g_, stage_addr + (uint32_t)((t * 16 + 4 * k) * 256));
      }
    }
  };
  // operand of table t, input block j: row n, chunk (4j + q) ^ n
  auto staged = [&](int t, int j) {
    return *reinterpret_cast<const f32x4*>(stage + (t * 16 + n) * 64 + 4 * ((4 * j + q) ^ n));
  };
  int stamp = 0;
  (void)stamp;
  STAMP(0, stamp++);
  issue_ids(wave0);
  {
    f32x4 wv[kFStagePer];
    int wdst[kFStagePer];
    float bv;
    stage_forward_load(T, wv, wdst, bv);
    float hw = 0.0f;
    if (threadIdx.x <= kHeadW) hw = F.wfold[threadIdx.x];
    issue_rows(wave0);                 // (waits for the ids alone: loads return in order)
    issue_ids(wave0 + nwaves);
    stage_forward_store(s_w, s_b, wv, wdst, bv);
    if (threadIdx.x <= kHeadW) s_hw[threadIdx.x] = hw;
  }
  __syncthreads();
  STAMP(0, stamp++);
  const float hc = s_hw[kHeadW];
  if (wave0 >= groups) return;
  // the first group's rows: everything issued so far (no stores yet).  The ids are "used" right behind every hand-placed
  // wait: the compiler then places ITS wait for them there, in straight-line code where it can count the stores behind
  // them, instead of a vmcnt(0) at the loop header (which is reached from two paths with different stores in flight)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" : "+v"(idu), "+v"(idi));
  for (int64_t g = wave0; g < groups; g += nwaves) {
    const int64_t row = g * 16 + n;
    const int64_t srow = row < m ? row : m;             // lanes without a sample store to the spare row m
    // staged rows of g -> operands
    f32x4 a0[4], xe[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 z = staged(0, j) + staged(1, j);
#pragma unroll
      for (int r = 0; r < 4; ++r) a0[j][r] = fmaxf(z[r], 0.0f);
    }
    // the head's extra columns 16q .. 16q+15 of this sample: chunks 4q + i
#pragma unroll
    for (int i = 0; i < 4; ++i)
      xe[i] = *reinterpret_cast<const f32x4*>(stage + (2 * 16 + n) * 64 + 4 * ((4 * q + i) ^ n)) *
              *reinterpret_cast<const f32x4*>(stage + (3 * 16 + n) * 64 + 4 * ((4 * q + i) ^ n));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage is read: it may be overwritten
    STAMP(0, stamp++);
    issue_rows(g + nwaves);
    issue_ids(g + 2 * nwaves);
    STAMP(0, stamp++);
    f32x4 y1[2], y2[1], y3[1];
    if constexpr (DBG & 2) {
      y1[0] = a0[0] + a0[2]; y1[1] = a0[1] + a0[3]; y2[0] = y1[0] * y1[1]; y3[0] = y2[0] + y1[0];
    } else {
      layer_fwd<0, 4>(s_w, s_b, lane, q, a0, y1);
      layer_fwd<1, 2>(s_w, s_b, lane, q, y1, y2);
      layer_fwd<2, 1>(s_w, s_b, lane, q, y2, y3);
    }
    STAMP(0, stamp++);
    if constexpr (!(DBG & 4)) {
#pragma unroll
      for (int b = 0; b < 2; ++b) stg4(T.y[0] + srow * T.ldy[0] + 16 * b + 4 * q, y1[b]);
      stg4(T.y[1] + srow * T.ldy[1] + 4 * q, y2[0]);
      stg4(T.y[2] + (q < 2 ? srow : m) * T.ldy[2] + 4 * (q & 1), y3[0]);
    }
    // head: prob = act([gmf | h] . wfold + cfold); the four lanes of a sample hold 16 + (q < 2 ? 4 : 0) terms each
    float dot = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(s_hw + 16 * q + 4 * i);
      dot = fmaf(xe[i][0], wv[0], dot); dot = fmaf(xe[i][1], wv[1], dot);
      dot = fmaf(xe[i][2], wv[2], dot); dot = fmaf(xe[i][3], wv[3], dot);
    }
    {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(s_hw + kP + 4 * (q & 1));
      float d2 = y3[0][0] * wv[0];
      d2 = fmaf(y3[0][1], wv[1], d2); d2 = fmaf(y3[0][2], wv[2], d2); d2 = fmaf(y3[0][3], wv[3], d2);
      dot += q < 2 ? d2 : 0.0f;
    }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    if constexpr (!(DBG & 16)) F.out[srow * F.ldout] = ctr_act(dot + hc, F.act);   // (the four lanes of a sample agree)
    else asm volatile("" ::"v"(dot));
    STAMP(0, stamp++);
    // the next group's rows and its successor's ids are older than this group's kFwdStores stores
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DBG & 4) ? ((DBG & 16) ? 0 : 1) : kFwdStores) : "memory");
    asm volatile("" : "+v"(idu), "+v"(idi));
  }
}

// ------------------------------------------------------------------ backward, per sample
// slab a workgroup leaves in the workspace, in the order its lanes hold the sums (coalesced 16-byte stores; the second
// pass, slab_reduce_role, does the index arithmetic once per output instead of every workgroup once per element):
//   [vector v][lane][register r]: dW accumulator vectors  v < 8: layer 0 block (v / 4, v % 4);  8, 9: layer 1;  10: layer 2
//   (register r of lane (q, lo) = row 16 b + 4 q + r, column 16 j + lo of its layer's weight gradient)
//   then kSmall sums: layer-0 bias (32), layer-1 bias (16), layer-2 bias (8), sum gz * h (8), sum gz (1)
// (the GMF part of the head's weight gradient comes from the table rows, ncfp_finish)
constexpr int slab_w(int l) {
  int o = 0;
  for (int i = 0; i < l; ++i) o += kN[i] * kK[i] + kN[i];
  return o;
}
constexpr int kSlabHead = slab_w(kL);                     // 2744 tower outputs: [dW_l | db_l] for the three layers
constexpr int kVecs = 8 + 2 + 1;                          // dW accumulator vectors of a lane
constexpr int kSmall = 32 + 16 + 8 + 8 + 1;               // bias sums, sum gz * h, sum gz
constexpr int kCopy = kVecs * 256 + 68;                   // one wave's sums parked in LDS (16-byte multiple)
constexpr int kSlab = kCopy;                              // 2884
constexpr int kStripP = 3 * kTile;                        // per wave: tiles A0 A1 | B0
// the staged operands of a sample group (floats, per wave): sixteen rows each of P_U, P_I, Y1, Y2, Y3 -- fetched
// coalesced by LDS-DMA, 16-byte chunks XOR-swizzled with the row so that both read patterns (sample-major ds_read_b128,
// unit-major ds_read_b32) spread over the banks
constexpr int kSgPU = 0, kSgPI = 1024, kSgY1 = 2048, kSgY2 = 2560, kSgY3 = 2816, kBwdStage = 3072;
constexpr int kBwdDma = 12;                               // row fetches per group
constexpr int kBwdStores = 10;                            // 8 bucket-row pieces + 2 slot records

struct Bwd {
  Ids ids;
  const float* ptab;
  const float* wfold;
  const float* prob; int64_t ldp;
  const float* gprob; int64_t ldgp;
  int act;
  const int32_t* counts;                       // (nu + ni) from the forward
  const int32_t* ranks;                        // (m + 1, 2)
  float* gz;                                   // (2m + 1, 64): buckets, user rows' slots first, one spare slot
  float* aux;                                  // (2m + 1, 4): {gz, partner id, row, -} per slot
  int32_t* offsets;                            // (nu + ni + 1): written by workgroup 0 for the later launches
  float* slabs;                                // (grid, kSlab)
  float* zero_a; int64_t zero_a_floats;        // cleared first: the segment sums (nu + ni, 128)
  float* zero_b; int64_t zero_b_floats;        // cleared first (nullable): the step's gradient buffer
};

__device__ __forceinline__ int sw1(int r) { return (r >> 1) & 7; }   // chunk swizzles of the 128 / 64 / 32-byte rows
__device__ __forceinline__ int sw2(int r) { return (r >> 2) & 3; }
__device__ __forceinline__ int sw3(int r) { return (r >> 3) & 1; }

__global__ void __launch_bounds__(kThreads)
ncfp_bwd_kernel(const Tower T, int64_t m, const Bwd B) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ __attribute__((aligned(16))) float s_hw[kHeadW + 4];
  __shared__ int s_scan[kWaves];
  float* s_wt = lds;
  const int lane = threadIdx.x & 63, q = lane >> 4, lo = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* tA = lds + kWFloats + wave * kStripP;
  float* tB = tA + 2 * kTile;
  float* stage = lds + kWFloats + kWaves * kStripP + wave * kBwdStage;
  const uint32_t stage_addr = lds_addr(stage);
  const int64_t nrows = B.ids.nu + B.ids.ni;
  int* s_off = reinterpret_cast<int*>(lds + kWFloats + kWaves * (kStripP + kBwdStage));   // nrows + 1 exclusive offsets
  const int64_t groups = (m + 15) / 16;
  const int64_t wave0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * kThreads) >> 6;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const float* pu = B.ptab;
  const float* pi = B.ptab + B.ids.nu * kN0;
  const uint32_t nu = (uint32_t)B.ids.nu, ni = (uint32_t)B.ids.ni;
  int stamp = 0;
  (void)stamp;
  STAMP(1, stamp++);

  // ---- clear what this call accumulates into (both 16-byte aligned multiples of 4 floats)
  {
    const int64_t t0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 4, step = (int64_t)gridDim.x * kThreads * 4;
    for (int64_t i = t0; i < B.zero_a_floats; i += step) stg4(B.zero_a + i, zero4);
    if (B.zero_b)
      for (int64_t i = t0; i < B.zero_b_floats; i += step) stg4(B.zero_b + i, zero4);
  }
  // ---- the fetch pipeline (see the note above ncfp_fwd_kernel).  Stage 1: the ids of this lane's sample (sample lo of
  // the group, the same in its four lanes).  Stage 2: the group's rows by LDS-DMA and the per-sample scalars.
  int64_t idu = 0, idi = 0;
  auto issue_ids = [&](int64_t g) {
    int64_t row = g * 16 + lo;
    row = row < m ? row : m - 1;
    idu = B.ids.uidx[row * B.ids.ustride];
    idi = B.ids.iidx[row * B.ids.istride];
  };
  struct Scal {
    float gp, pb;
    int ru, ri;                       // ranks of this lane's sample in its user / item row
    uint32_t u, i;                    // its ids, clamped (bad ids: row 0, no slot)
    bool ubad, ibad;
  };
  auto issue_rows = [&](int64_t g, Scal& r) {
    int64_t rc = g * 16 + lo;
    rc = rc < m ? rc : m - 1;
    r.ubad = (uint64_t)idu >= nu;
    r.ibad = (uint64_t)idi >= ni;
    r.u = r.ubad ? 0u : (uint32_t)idu;
    r.i = r.ibad ? 0u : (uint32_t)idi;
    // P_U / P_I: instruction k moves rows 4k .. 4k+3, lane l chunk (l % 16) ^ row of row 4k + l / 16
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = 4 * k + q;
      const int src = (lane & 48) | row;                   // a lane of this quarter that holds sample row's ids
      const uint32_t ur = (uint32_t)__shfl((int)r.u, src, 64), ir = (uint32_t)__shfl((int)r.i, src, 64);
      const uint32_t col = 4u * (uint32_t)(lo ^ row);
      dma16(pu + ur * 64u + col, stage_addr + (uint32_t)((kSgPU + 4 * k * 64) * 4));
      dma16(pi + ir * 64u + col, stage_addr + (uint32_t)((kSgPI + 4 * k * 64) * 4));
    }
    {
      // Y1 (32 floats a row): instruction k moves rows 8k .. 8k+7, lane l chunk (l % 8) ^ sw1(row) of row 8k + l / 8
      const int64_t g16 = g * 16;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int row = 8 * k + (lane >> 3);
        int64_t gr = g16 + row;
        gr = gr < m ? gr : m - 1;
        dma16(T.y[0] + gr * T.ldy[0] + 4 * ((lane & 7) ^ sw1(row)), stage_addr + (uint32_t)((kSgY1 + 8 * k * 32) * 4));
      }
      {   // Y2 (16 floats): lane l chunk (l % 4) ^ sw2(row) of row l / 4
        const int row = lane >> 2;
        int64_t gr = g16 + row;
        gr = gr < m ? gr : m - 1;
        dma16(T.y[1] + gr * T.ldy[1] + 4 * ((lane & 3) ^ sw2(row)), stage_addr + (uint32_t)(kSgY2 * 4));
      }
      {   // Y3 (8 floats): lane l chunk (l % 2) ^ sw3(row) of row (l / 2) % 16 (the upper half-wave repeats the lower)
        const int row = (lane >> 1) & 15;
        int64_t gr = g16 + row;
        gr = gr < m ? gr : m - 1;
        dma16(T.y[2] + gr * T.ldy[2] + 4 * ((lane & 1) ^ sw3(row)), stage_addr + (uint32_t)(kSgY3 * 4));
      }
    }
    r.gp = B.gprob[rc * B.ldgp];
    r.pb = B.prob[rc * B.ldp];
    r.ru = B.ranks[2 * rc];
    r.ri = B.ranks[2 * rc + 1];
  };
  Scal sc;
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
  issue_ids(wave0);
  {
    f32x4 wv[kStagePer];
    int wdst[kStagePer];
    stage_transposed_load(T, wv, wdst);
    const float hw = threadIdx.x <= kHeadW ? B.wfold[threadIdx.x] : 0.0f;
    // the per-row sample counts into LDS, kCnt coalesced loads in flight per thread (rows past the end re-read the
    // last): ml-100k's 2625 rows are ONE round trip (four in flight were three)
    constexpr int kCnt = 12;
    for (int64_t base = 0; base < nrows; base += kCnt * kThreads) {
      int v[kCnt];
#pragma unroll
      for (int e = 0; e < kCnt; ++e) {
        const int64_t i = base + e * kThreads + threadIdx.x;
        v[e] = B.counts[(i < nrows ? i : nrows - 1) * kCountStride];
      }
#pragma unroll
      for (int e = 0; e < kCnt; ++e) {
        const int64_t i = base + e * kThreads + threadIdx.x;
        if (i < nrows) s_off[i] = v[e];
      }
    }
    issue_rows(wave0, sc);
    issue_ids(wave0 + nwaves);
    stage_transposed_store(s_wt, wv, wdst);
    if (threadIdx.x <= kHeadW) s_hw[threadIdx.x] = hw;
  }
  __syncthreads();
  STAMP(1, stamp++);   // weights staged, counts in LDS, first group requested
  // ---- exclusive scan of the counts (users, then items), in place: every workgroup for itself
  {
    const int per = (int)((nrows + kThreads - 1) / kThreads);
    const int64_t i0 = (int64_t)threadIdx.x * per;
    int sum = 0;
    for (int e = 0; e < per; ++e)
      if (i0 + e < nrows) sum += s_off[i0 + e];
    int inc = sum;   // inclusive scan of `sum` over the workgroup's threads
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
        const int c = s_off[i0 + e];
        s_off[i0 + e] = run;
        run += c;
      }
    if (threadIdx.x == kThreads - 1) s_off[nrows] = base + inc;
    __syncthreads();
    if (blockIdx.x == 0)
      for (int64_t i = threadIdx.x; i <= nrows; i += kThreads) B.offsets[i] = s_off[i];
  }
  STAMP(1, stamp++);   // scan done
  // the first group's rows and scalars, the second group's ids (nothing else is in flight but workgroup 0's offsets)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" : "+v"(idu), "+v"(idi), "+v"(sc.gp), "+v"(sc.pb), "+v"(sc.ru), "+v"(sc.ri));

  for (int64_t g = wave0; g < groups; g += nwaves) {
    const bool live = g * 16 + lo < m;
    // ---- staged rows of g -> operands
    const float gp = live ? sc.gp : 0.0f;                    // a dead lane's gz is zero: it adds nothing anywhere
    const float pb = sc.pb;
    f32x4 y3d, y2d, y2t, y1d[2], y1t[2], a0d[4], a0t[4];
    y3d = *reinterpret_cast<const f32x4*>(stage + kSgY3 + lo * 8 + 4 * ((q & 1) ^ sw3(lo)));
    y3d = q < 2 ? y3d : zero4;
    y2d = *reinterpret_cast<const f32x4*>(stage + kSgY2 + lo * 16 + 4 * (q ^ sw2(lo)));
#pragma unroll
    for (int b = 0; b < 2; ++b) y1d[b] = *reinterpret_cast<const f32x4*>(stage + kSgY1 + lo * 32 + 4 * ((4 * b + q) ^ sw1(lo)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int at = lo * 64 + 4 * ((4 * j + q) ^ lo);
      const f32x4 z = *reinterpret_cast<const f32x4*>(stage + kSgPU + at) + *reinterpret_cast<const f32x4*>(stage + kSgPI + at);
#pragma unroll
      for (int r = 0; r < 4; ++r) a0d[j][r] = fmaxf(z[r], 0.0f);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = 4 * q + c, ch = lo >> 2, w = lo & 3;
      y2t[c] = stage[kSgY2 + row * 16 + 4 * (ch ^ sw2(row)) + w];
#pragma unroll
      for (int b = 0; b < 2; ++b) y1t[b][c] = stage[kSgY1 + row * 32 + 4 * ((4 * b + ch) ^ sw1(row)) + w];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int at = row * 64 + 4 * ((4 * j + ch) ^ row) + w;
        a0t[j][c] = fmaxf(stage[kSgPU + at] + stage[kSgPI + at], 0.0f);
      }
    }
    const int uu = (int)sc.u, ii = (int)sc.i;
    const int su = (live && !sc.ubad && sc.ru >= 0) ? sc.ru + s_off[uu] : -1;
    const int si = (live && !sc.ibad && sc.ri >= 0) ? sc.ri + s_off[nu + ii] : -1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the stage is read: it may be overwritten
    STAMP(1, stamp++);   // operands read
    issue_rows(g + nwaves, sc);
    issue_ids(g + 2 * nwaves);
    STAMP(1, stamp++);   // next group requested
    // ---- head: gz, the head's sums, the tower's (masked) gY
    const float gzs = gp * ctr_act_grad(pb, B.act);
    if (q == 0) hc += gzs;
    f32x4 gz3[1];
    {
      const f32x4 wv = q < 2 ? *reinterpret_cast<const f32x4*>(s_hw + kP + 4 * q) : zero4;
      hy += gzs * y3d;                                        // (y3d is zero for q >= 2)
      gz3[0] = relu_mask(gzs * wv, y3d);
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
      dx_layer<2, 1>(s_wt, lane, gz3, w0, w1, [&](int, const f32x4& d0, const f32x4&) { gz2[0] = relu_mask(d0, y2d); });
      sb1 += gz2[0];
      tiles_put<1>(tB, q, lo, gz2);
      dx_first<1>(s_wt, lane, w0, w1);
#pragma unroll
      for (int c = 0; c < 4; ++c) dw2 = __builtin_amdgcn_mfma_f32_16x16x4f32(tg[0][c], y2t[c], dw2, 0, 0, 0);
    }
    // ---- layer 1 (32 -> 16)
    f32x4 gz1[2];
    {
      f32x4 tg[1];
      tiles_get<1>(tB, q, lo, tg);
      dx_layer<1, 1>(s_wt, lane, gz2, w0, w1, [&](int, const f32x4& d0, const f32x4& d1) {
        gz1[0] = relu_mask(d0, y1d[0]);
        gz1[1] = relu_mask(d1, y1d[1]);
      });
#pragma unroll
      for (int b = 0; b < 2; ++b) sb0[b] += gz1[b];
      tiles_put<2>(tA, q, lo, gz1);
      dx_first<0>(s_wt, lane, w0, w1);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) dw1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(tg[0][c], y1t[j][c], dw1[j], 0, 0, 0);
    }
    STAMP(1, stamp++);   // head + layers 2, 1 done
    // ---- layer 0 (64 -> 32): its input gradient, masked by relu'(z0), IS gz0 -- stored into the sample's two buckets
    {
      f32x4 tg[2];
      tiles_get<2>(tA, q, lo, tg);
      // a sample without a slot (bad id, padding lane) writes to the spare slot behind the buckets
      float* du = B.gz + (int64_t)(su >= 0 ? su : 2 * m) * kN0 + 4 * q;
      float* di = B.gz + (int64_t)(si >= 0 ? si : 2 * m) * kN0 + 4 * q;
      dx_layer<0, 2>(s_wt, lane, gz1, w0, w1, [&](int j, const f32x4& d0, const f32x4& d1) {
        const f32x4 g0 = relu_mask(d0, a0d[j]), g1 = relu_mask(d1, a0d[j + 1]);
        stg4(du + 16 * j, g0);
        stg4(du + 16 * (j + 1), g1);
        stg4(di + 16 * j, g0);
        stg4(di + 16 * (j + 1), g1);
      });
      // (the four lanes of a sample write the same record: an unconditional, countable store)
      stg4(B.aux + (int64_t)(su >= 0 ? su : 2 * m) * 4, f32x4{gzs, __int_as_float(ii), __int_as_float(uu), 0.0f});
      stg4(B.aux + (int64_t)(si >= 0 ? si : 2 * m) * 4, f32x4{gzs, __int_as_float(uu), __int_as_float((int)nu + ii), 0.0f});
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            dw0[b][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(tg[b][c], a0t[j][c], dw0[b][j], 0, 0, 0);
    }
    STAMP(1, stamp++);   // layer 0 done
    // the next group's rows, scalars and its successor's ids are older than this group's kBwdStores stores
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kBwdStores) : "memory");
    asm volatile("" : "+v"(idu), "+v"(idi), "+v"(sc.gp), "+v"(sc.pb), "+v"(sc.ru), "+v"(sc.ri));
  }

  // ---- the workgroup's partial: every wave parks its sums (weights, tiles and stages are dead), the four copies are
  // summed on the way out
  STAMP(1, stamp++);   // loop end
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (row fetches requested for a group past the end have landed)
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
    if (lane == 0) {
      small[64] = vc;
      small[65] = small[66] = small[67] = 0.0f;
    }
  }
  __syncthreads();
  float* out = B.slabs + (int64_t)blockIdx.x * kSlab;
  for (int i = threadIdx.x; i < kCopy / 4; i += kThreads) {
    const f32x4 t = (*reinterpret_cast<const f32x4*>(lds + 4 * i) + *reinterpret_cast<const f32x4*>(lds + kCopy + 4 * i)) +
                    (*reinterpret_cast<const f32x4*>(lds + 2 * kCopy + 4 * i) + *reinterpret_cast<const f32x4*>(lds + 3 * kCopy + 4 * i));
    stg4(out + 4 * i, t);
  }
  STAMP(1, stamp++);   // slab written
}

// ------------------------------------------------------------------ segment sums over the buckets
// ST[v] = [ S[v] (64) | T[v] (64) ],  S[v] = sum of the gz0 rows in row v's bucket,  T[v] = sum gz_b * partner row.
// A lane group of sixteen owns sixteen consecutive SLOTS, whatever rows they belong to: equal work per wave under any
// id distribution.  Every load of the group is requested before any is consumed -- the slot records first (one round
// trip), then the sixteen bucket rows and the sixteen partner rows they name (a second one): a load-use loop over the
// slots was eight dependent round trips, 15 us for a kernel that moves 50 MB.  The group keeps running sums for the row
// it is in and adds them to ST when the row changes (and at its end): one 64-byte atomic segment per sixteen lanes and
// quarter row, after a transposition through LDS (lane lo of the loads holds columns 4lo .. 4lo+3; an atomic
// instruction wants sixteen consecutive floats from sixteen lanes).
//
// The same launch carries, on workgroups of their own, the second pass over ncfp_bwd's slabs (tower dW / db: 32
// outputs x 8 part-lanes per workgroup, fixed order) and the head fold's chain rule from the slabs' head sums
// (ctr_fold_head_bwd's map without the GMF part): both only depend on ncfp_bwd, and run beside the segment sums
// instead of in a launch of their own in front of them.
struct Seg {
  const float* gz; const float* aux; const int32_t* offsets;
  const float* gmf_u; const float* gmf_i;
  int64_t nu, ni;
  float* st;                                   // (nu + ni, 128), zeroed
  int seg_blocks, red_blocks;                  // roles by blockIdx.x: [0, seg) segment sums, [seg, seg + red) slabs, last: fold
  const float* slabs; int parts;
  float* gw[kL]; float* gb[kL];                // tower gradients (+=)
  // head fold chain rule: u = linear2.weight (128), w / b = `linear` (64 x 8)
  const float* fold_u; const float* fold_w; int64_t fold_ldw; const float* fold_b;
  float* g_u; float* g_w; int64_t ld_g_w; float* g_b; float* g_b2;   // (+=), nullable
};

// float offset inside a slab of tower output e (e in the order [dW_l | db_l], l = 0..2)
__device__ __forceinline__ int slab_raw_of(int e) {
  int l = 0, r = e;
#pragma unroll
  for (int i = 0; i < kL; ++i)
    if (e >= slab_w(i)) { l = i; r = e - slab_w(i); }
  const int K = l == 0 ? kK[0] : l == 1 ? kK[1] : kK[2], N = l == 0 ? kN[0] : l == 1 ? kN[1] : kN[2];
  if (r >= N * K) return kVecs * 256 + (l == 0 ? 0 : l == 1 ? 32 : 48) + (r - N * K);   // bias sums
  const int row = r / K, col = r - row * K;
  const int J = K / 16, v0 = l == 0 ? 0 : l == 1 ? 8 : 10;
  const int v = v0 + (row >> 4) * J + (col >> 4);
  return (v * 64 + ((row >> 2) & 3) * 16 + (col & 15)) * 4 + (row & 3);
}

__device__ __forceinline__ void slab_reduce_role(const Seg& A, int blk) {
  // outputs [32 blk, 32 blk + 32) of the kSlabHead tower sums: lane (o, pl) adds every 8th partial, 8 loads in flight
  __shared__ float s_part[kWaves][32];
  const int o = threadIdx.x & 31, pl = threadIdx.x >> 5, wave = threadIdx.x >> 6;
  const int e = blk * 32 + o;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (e < kSlabHead) {
    const float* src = A.slabs + slab_raw_of(e);
    int p = pl;
    for (; p + 7 * 8 < A.parts; p += 8 * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += src[(int64_t)(p + 8 * u) * kSlab];
    }
    for (; p < A.parts; p += 8) acc[0] += src[(int64_t)p * kSlab];
  }
  float t = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  t += __shfl_xor(t, 32, 64);
  if ((threadIdx.x & 63) < 32) s_part[wave][o] = t;
  __syncthreads();
  if (threadIdx.x < 32 && e < kSlabHead) {
    const float total = (s_part[0][o] + s_part[1][o]) + (s_part[2][o] + s_part[3][o]);
    int l = 0, r = e;
#pragma unroll
    for (int i = 0; i < kL; ++i)
      if (e >= slab_w(i)) { l = i; r = e - slab_w(i); }
    const int wn = l == 0 ? kN[0] * kK[0] : l == 1 ? kN[1] * kK[1] : kN[2] * kK[2];
    float* dst = r < wn ? A.gw[l] + r : A.gb[l] + (r - wn);
    dst[0] += total;
  }
}

__device__ __forceinline__ void head_fold_role(const Seg& A) {
  // nine columns (sum gz * h [8], sum gz) x 28 part-lanes: every load of a thread in flight at once
  __shared__ float s_p[28][12];
  __shared__ float s_g[12];
  const int col = threadIdx.x % 9, pl = threadIdx.x / 9;
  float acc = 0.0f;
  if (pl < 28) {
    const float* src = A.slabs + kVecs * 256 + 56 + col;
    float v[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) v[u] = 0.0f;
    for (int p0 = pl; p0 < A.parts; p0 += 28 * 10) {
#pragma unroll
      for (int u = 0; u < 10; ++u)
        if (p0 + 28 * u < A.parts) v[u] += src[(int64_t)(p0 + 28 * u) * kSlab];
    }
#pragma unroll
    for (int u = 0; u < 10; ++u) acc += v[u];
    s_p[pl][col] = acc;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    float t = 0.0f;
    for (int i = 0; i < 28; ++i) t += s_p[i][threadIdx.x];
    s_g[threadIdx.x] = t;
  }
  __syncthreads();
  const float* u = A.fold_u + kP;
  const float gc = s_g[kNL];
  for (int t = threadIdx.x; t < 64 * kNL; t += kThreads) {            // g `linear`.weight[i][q] += u[64 + i] * hy[q]
    const int i = t / kNL, q = t % kNL;
    if (A.g_w) A.g_w[(int64_t)i * A.ld_g_w + q] += u[i] * s_g[q];
  }
  if (threadIdx.x < 64) {
    const int i = threadIdx.x;
    if (A.g_u) {                                                      // g linear2.weight[64 + i]
      float sacc = A.fold_b ? A.fold_b[i] * gc : 0.0f;
#pragma unroll
      for (int q = 0; q < kNL; ++q) sacc = fmaf(A.fold_w[(int64_t)i * A.fold_ldw + q], s_g[q], sacc);
      A.g_u[kP + i] += sacc;
    }
    if (A.g_b) A.g_b[i] += u[i] * gc;
  }
  if (threadIdx.x == 64 && A.g_b2) A.g_b2[0] += gc;
}

constexpr int kSegRange = 32;   // consecutive slots a lane group of sixteen owns
constexpr int kSegBatch = 8;    // slots whose loads are in flight together

__global__ void __launch_bounds__(kThreads)
ncfp_segsum_kernel(const Seg A) {
  if ((int)blockIdx.x >= A.seg_blocks) {
    if ((int)blockIdx.x < A.seg_blocks + A.red_blocks) slab_reduce_role(A, (int)blockIdx.x - A.seg_blocks);
    else head_fold_role(A);
    return;
  }
  // A lane group owns kSegRange consecutive slots and walks them kSegBatch at a time, software-pipelined: while batch k
  // is summed, the rows of batch k + 1 (bucket row + the partner row its record names) and the records of batch k + 2
  // are in flight.  A row whose whole bucket lies inside the range is STORED (nobody else adds to it); only the first and
  // the last row of a range can continue in a neighbour's and are added atomically -- 2 per 64 slots instead of one per
  // 12: the first version flushed every run with atomics and ran at the memory side's atomic rate (240 K 64-byte
  // requests, TCC_EA0_ATOMIC), not at the rate the buckets can be read.
  __shared__ __attribute__((aligned(16))) float s_t[kThreads / 16][2][68];
  const int lane = threadIdx.x & 63, lo = lane & 15, grp = threadIdx.x >> 4;
  const int64_t total = A.offsets[A.nu + A.ni];
  const int64_t s0 = ((int64_t)blockIdx.x * (kThreads / 16) + grp) * kSegRange;
  const int64_t s1 = s0 + kSegRange < total ? s0 + kSegRange : total;   // (may be <= s0: nothing to do, but stay for the shuffles)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 accs = zero4, acct = zero4;
  int cur = -1;
  bool cur_inside = false;   // the run being summed began inside this range (its row's bucket does not start earlier)
  auto flush = [&](int v, bool complete) {
    // a row whose whole bucket lies inside [s0, s1): plain stores; else atomics (transposed through LDS to 64-byte segments)
    float* dst = A.st + (int64_t)v * 128;
    if (complete) {
      stg4(dst + 4 * lo, accs);
      stg4(dst + 64 + 4 * lo, acct);
    } else {
      float* t = &s_t[grp][0][0];
      *reinterpret_cast<f32x4*>(t + 4 * lo) = accs;
      *reinterpret_cast<f32x4*>(t + 68 + 4 * lo) = acct;
      asm volatile("" ::: "memory");   // (same lane group, same wave: the LDS pipe is in order)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ctr_atomic_add_global(dst + 16 * r + lo, t[16 * r + lo]);
        ctr_atomic_add_global(dst + 64 + 16 * r + lo, t[68 + 16 * r + lo]);
      }
      asm volatile("" ::: "memory");
    }
  };
  const int64_t last = total > 0 ? total - 1 : 0;
  // the rows of the slots just outside the range (requested with the first records): does the first / last run continue?
  const f32x4 rec_before = ldg4(A.aux + (s0 > 0 ? (s0 - 1 < total ? s0 - 1 : last) : 0) * 4);
  const f32x4 rec_after = ldg4(A.aux + (s1 < total ? s1 : last) * 4);
  const int v_before = s0 > 0 ? __float_as_int(rec_before[2]) : -1;
  const int v_after = s1 < total ? __float_as_int(rec_after[2]) : -1;
  auto slot = [&](int64_t s) { return s < s1 ? s : (s1 > s0 ? s1 - 1 : last); };   // clamped: always a written slot
  auto load_aux = [&](int64_t base, f32x4 (&ax)[kSegBatch]) {
#pragma unroll
    for (int k = 0; k < kSegBatch; ++k) ax[k] = ldg4(A.aux + slot(base + k) * 4);
  };
  auto load_rows = [&](int64_t base, const f32x4 (&ax)[kSegBatch], f32x4 (&g)[kSegBatch], f32x4 (&p)[kSegBatch]) {
#pragma unroll
    for (int k = 0; k < kSegBatch; ++k) {
      const int pid = __float_as_int(ax[k][1]), v = __float_as_int(ax[k][2]);
      g[k] = ldg4(A.gz + slot(base + k) * kN0 + 4 * lo);
      p[k] = ldg4((v < A.nu ? A.gmf_i : A.gmf_u) + (int64_t)pid * kP + 4 * lo);
    }
  };
  auto sum = [&](int64_t base, const f32x4 (&ax)[kSegBatch], const f32x4 (&g)[kSegBatch], const f32x4 (&p)[kSegBatch]) {
#pragma unroll
    for (int k = 0; k < kSegBatch; ++k) {
      if (base + k < s1) {
        const int v = __float_as_int(ax[k][2]);
        if (v != cur) {
          if (cur >= 0) flush(cur, cur_inside);       // it ended here, inside the range
          cur_inside = base + k > s0 || v != v_before;
          cur = v;
          accs = acct = zero4;
        }
        accs += g[k];
        acct += ax[k][0] * p[k];
      }
    }
  };
  if (total <= 0) return;
  // batch k: records axA, rows (gA, pA); batch k + 1: records axB
  f32x4 axA[kSegBatch], axB[kSegBatch], axC[kSegBatch], gA[kSegBatch], pA[kSegBatch], gB[kSegBatch], pB[kSegBatch];
  load_aux(s0, axA);
  load_rows(s0, axA, gA, pA);
  load_aux(s0 + kSegBatch, axB);
  constexpr int kPairs = kSegRange / (2 * kSegBatch);
#pragma unroll 1
  for (int it = 0; it < kPairs; ++it) {
    const int64_t base = s0 + (int64_t)it * 2 * kSegBatch;
    load_rows(base + kSegBatch, axB, gB, pB);          // rows of k + 1 (waits for its records)
    load_aux(base + 2 * kSegBatch, axC);               // records of k + 2
    sum(base, axA, gA, pA);                            // k (its rows were requested a batch ago)
    load_rows(base + 2 * kSegBatch, axC, gA, pA);      // rows of k + 2
    load_aux(base + 3 * kSegBatch, axA);               // records of k + 3
    sum(base + kSegBatch, axB, gB, pB);                // k + 1
#pragma unroll
    for (int k = 0; k < kSegBatch; ++k) {              // next pair: k + 2 in (axA, gA, pA), records of k + 3 in axB
      const f32x4 t = axA[k];
      axA[k] = axC[k];
      axB[k] = t;
    }
  }
  if (cur >= 0) flush(cur, cur_inside && cur != v_after);
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
  int32_t* counts; int64_t ncounts;                                  // the step's sample counters: left all zero for the next forward
};

__global__ void __launch_bounds__(kThreads)
ncfp_finish_kernel(const Fin A) {
  __shared__ __attribute__((aligned(16))) float s_dw[2][16 * 256];
  __shared__ float s_sm[kWaves][2][64];
  const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15, wave = threadIdx.x >> 6;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < A.ncounts; i += (int64_t)gridDim.x * kThreads)
    A.counts[i] = 0;    // (every reader -- the per-sample backward's scan -- is two launches back)
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
    // every operand is requested before any is consumed: they are independent of each other, and a load-use order was
    // eight dependent round trips on a handful of workgroups (12 us for 40 MFLOP)
    // ---- sample-major operands: this lane's row n, columns 16j + 4q ..
    // (unconditional loads from a clamped row, zeroed afterwards: a load inside a conditional is a round trip of its own)
    f32x4 sd[4], td[4], gd[4], og[4], ot[4], wf[4];
    const int64_t crow = live ? row : rows - 1;
    const float* ogp = ggmf ? ggmf : gmf;      // (a missing gradient buffer: read something valid, the store is skipped)
    const float* otp = gtab ? gtab : tab;
    const float keep = live ? 1.0f : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sd[j] = ldg4(st + crow * 128 + 16 * j + 4 * q);
      td[j] = ldg4(st + crow * 128 + 64 + 16 * j + 4 * q);
      gd[j] = ldg4(gmf + crow * kP + 16 * j + 4 * q);
      og[j] = ldg4(ogp + crow * kP + 16 * j + 4 * q);
      ot[j] = ldg4(otp + crow * kH + 16 * j + 4 * q);
      wf[j] = ldg4(A.wfold + 16 * j + 4 * q);
    }
    // ---- unit-major operands: unit / input n of rows 4q + c;  A operands of dMLP: W0[16j + 4q + c][coff + 16b + n]
    f32x4 stt[4], xt[4], wt[4][4];
    float okc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t r = r0 + 4 * q + c;
      const int64_t rr = r < rows ? r : rows - 1;
      okc[c] = r < rows ? 1.0f : 0.0f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        stt[b][c] = st[rr * 128 + 16 * b + n];
        xt[b][c] = tab[rr * kH + 16 * b + n];
#pragma unroll
        for (int j = 0; j < 4; ++j) wt[j][b][c] = A.w0[(int64_t)(16 * j + 4 * q + c) * A.ldw0 + coff + 16 * b + n];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sd[j] *= keep;
      td[j] *= keep;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) stt[b][c] *= okc[c];
    // dGMF[row] += wfold[:64] * T[row];   sum_rows GMF * T
    if (ggmf && live) {
#pragma unroll
      for (int j = 0; j < 4; ++j) stg4(ggmf + row * kP + 16 * j + 4 * q, og[j] + wf[j] * td[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gw[j] = gd[j] * td[j];
      colsum[j] = sd[j];
    }
    // ---- dMLP^T (64 inputs x 16 rows) = W0half^T (64 x 64 units) . S^T (64 units x 16 rows)
    if (gtab) {
      f32x4 acc[4] = {ot[0], ot[1], ot[2], ot[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[j][b][c], sd[j][c], acc[b], 0, 0, 0);
      if (live) {
#pragma unroll
        for (int b = 0; b < 4; ++b) stg4(gtab + row * kH + 16 * b + 4 * q, acc[b]);
      }
    }
    // ---- dW0half (64 units x 64 inputs) += S^T . X
    if (A.g_w0) {
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

#ifdef CTR_STAMPS
extern "C" __attribute__((visibility("default"))) int ctr_ncfp_debug_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 128);
}
#endif

static int64_t workspace_floats(int64_t batch, int64_t num_users, int64_t num_items) {
  const int64_t rows = num_users + num_items;
  const int64_t groups = ctr_ceil_div(batch > 0 ? batch : 1, 16);
  int64_t grid = ctr_ceil_div(groups, kWaves);
  if (grid > 768) grid = 768;
  // buckets (2B + 1, 64) | slot records (2B + 1, 4) | segment sums (rows, 128) | offsets (rows + 1) | slabs
  // (the spare slot 2B takes the stores of samples without a slot: bad ids, the padding lanes of the last group)
  return (2 * batch + 1) * kN0 + (2 * batch + 1) * 4 + rows * 128 + (rows + 1 + 3) / 4 * 4 + grid * (int64_t)kSlab;
}

extern "C" int ctr_ncf_proj_workspace_floats(int64_t batch, int64_t num_users, int64_t num_items, int64_t* floats) {
  CTR_REQUIRE(floats && batch >= 0 && num_users >= 0 && num_items >= 0, CTR_EINVAL);
  *floats = workspace_floats(batch, num_users, num_items);
  return CTR_OK;
}

// `phases`: 0 = the whole call; else a mask of its launches (a profiler brackets them one by one): forward 1 = projected
// tables + head fold, 2 = the per-sample kernel; backward 1 = the per-sample kernel, 2 = segment sums + slab reduction +
// head fold, 4 = the table-row products
extern "C" int ctr_ncf_proj_fwd(const ctr_ncf_proj_t* d, void* stream) {
  CTR_REQUIRE(d && d->batch >= 0, CTR_EINVAL);
  const int phases = d->phases ? d->phases : 3;
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
  const Ids ids{d->user_idx, d->user_stride, d->item_idx, d->item_stride, nu, ni};
  // training: the samples' ranks, part in this launch, part in the next (RankJob)
  static const int rank_split = [] { const char* e = getenv("CTR_NCFP_RANK_SPLIT"); return e ? atoi(e) : 50; }();   // per cent (sweep: dev/r03_rank_split.sh)
  const int64_t m_first = d->training ? d->batch * rank_split / 100 / kThreads * kThreads : 0;
  const int proj_blocks = (int)ctr_ceil_div(pwaves, kWaves) + 1;
  int64_t rank_a = ctr_ceil_div(m_first, kThreads);
  if (rank_a > 256) rank_a = 256;
  const Prep P{d->mlp_user, d->mlp_item, d->layers[0].w, d->layers[0].k, d->layers[0].b, d->ptab, nu, ni,
               d->head_w, d->proj_w, d->ld_proj_w, d->proj_b, d->head_b, d->wfold,
               RankJob{ids, d->counts, d->ranks, 0, m_first, proj_blocks}};
  if (phases & 1) hipLaunchKernelGGL(ncfp_prep_kernel, dim3((unsigned)(proj_blocks + rank_a)), dim3(kThreads), 0, st, P);
  rc = ctr_launch_status();
  if (rc != CTR_OK || d->batch == 0 || !(phases & 2)) return rc;
  const int64_t groups = ctr_ceil_div(d->batch, 16);
  int64_t grid = ctr_ceil_div(groups, kWaves);
  // one workgroup per CU, every wave walks several groups: 25.6 us at batch 65536 against 28 us with two per CU
  static const int fwd_wgs = [] { const char* e = getenv("CTR_NCFP_FWD_WGS"); return e ? atoi(e) : 256; }();
  if (grid > fwd_wgs) grid = fwd_wgs;
  // rank workgroups (training): one sample per thread up to a chip's worth of them, behind the per-sample ones
  int64_t rank_blocks = d->training ? ctr_ceil_div(d->batch - m_first, kThreads) : 0;
  if (rank_blocks > 256) rank_blocks = 256;
  const Fwd F{ids, d->ptab, d->gmf_user, d->gmf_item, d->wfold, d->prob, d->ldprob, d->head_act, d->err_flag,
              RankJob{ids, d->counts, d->ranks, m_first, d->training ? d->batch : m_first, (int)grid}};
  static const int dbg = [] { const char* e = getenv("CTR_NCFP_DBG"); return e ? atoi(e) : 0; }();
  constexpr size_t fwd_lds = sizeof(float) * (kWFloats + kBFloats + 80 + kWaves * kFwdStage);
  switch (dbg) {
#define CTR_DBG_CASE(V)                                                                                                  \
  case V:                                                                                                                \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(ncfp_fwd_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)fwd_lds) != hipSuccess)                                                                 \
      return CTR_ELAUNCH;                                                                                                \
    hipLaunchKernelGGL(ncfp_fwd_kernel<V>, dim3((unsigned)(grid + rank_blocks)), dim3(kThreads), fwd_lds, st, T, d->batch, F); \
    break
    CTR_DBG_CASE(54); CTR_DBG_CASE(62); CTR_DBG_CASE(8);
    default: CTR_DBG_CASE(0);
#undef CTR_DBG_CASE
  }
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
  float* gzb = ws;            ws += (2 * m + 1) * kN0;
  float* aux = ws;            ws += (2 * m + 1) * 4;
  float* stt = ws;            ws += rows * 128;
  int32_t* offs = reinterpret_cast<int32_t*>(ws); ws += (rows + 1 + 3) / 4 * 4;
  float* slabs = ws;
  const int64_t groups = ctr_ceil_div(m, 16);
  int64_t grid = ctr_ceil_div(groups, kWaves);
  static const int bwd_wgs = [] { const char* e = getenv("CTR_NCFP_BWD_WGS"); return e ? atoi(e) : 256; }();
  if (grid > bwd_wgs) grid = bwd_wgs;
  const Ids ids{d->user_idx, d->user_stride, d->item_idx, d->item_stride, nu, ni};
  const Bwd B{ids, d->ptab, d->wfold, d->prob, d->ldprob, g->gprob, g->ldgprob, d->head_act, d->counts, d->ranks, gzb, aux,
              offs, slabs, stt, rows * 128, g->zero_buf, g->zero_buf ? g->zero_floats : 0};
  const int64_t main_f = kWFloats + kWaves * (kStripP + kBwdStage) + (rows + 1 + 3) / 4 * 4, copy_f = (int64_t)kWaves * kCopy;
  const size_t lds_bytes = sizeof(float) * (size_t)(main_f > copy_f ? main_f : copy_f);
  CTR_REQUIRE(lds_bytes <= 150 * 1024, CTR_ELIMIT);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(ncfp_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds_bytes) != hipSuccess)
    return CTR_ELAUNCH;
  const int phases = g->phases ? g->phases : 7;
  if (phases & 1) hipLaunchKernelGGL(ncfp_bwd_kernel, dim3((unsigned)grid), dim3(kThreads), lds_bytes, st, T, m, B);
  rc = ctr_launch_status();
  if (rc != CTR_OK) return rc;
  // segment sums over the buckets, and beside them (own workgroups) the tower's dW / db partials and the head fold's
  // chain rule (its GMF part arrives from ncfp_finish)
  Seg S{gzb, aux, offs, d->gmf_user, d->gmf_item, nu, ni, stt, (int)ctr_ceil_div(2 * m, (kThreads / 16) * kSegRange),
        (kSlabHead + 31) / 32,
        slabs, (int)grid, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, d->head_w, d->proj_w, d->ld_proj_w,
        d->proj_b, g->g_head_w, g->g_proj_w, g->ld_g_proj_w, g->g_proj_b, g->g_head_b};
  for (int l = 0; l < kL; ++l) {
    S.gw[l] = g->layers[l + 1].gw;
    S.gb[l] = g->layers[l + 1].gb;
  }
  {
    // timing experiments (results are wrong): CTR_NCFP_SEG_DBG bit 0 drops the segment-sum workgroups, bit 1 the slab ones
    static const int sdbg = [] { const char* e = getenv("CTR_NCFP_SEG_DBG"); return e ? atoi(e) : 0; }();
    if (sdbg & 1) S.seg_blocks = 0;
    if (sdbg & 2) S.red_blocks = 0;
  }
  if (phases & 2)
    hipLaunchKernelGGL(ncfp_segsum_kernel, dim3((unsigned)(S.seg_blocks + S.red_blocks + 1)), dim3(kThreads), 0, st, S);
  rc = ctr_launch_status();
  if (rc != CTR_OK) return rc;
  const Fin N{stt, d->mlp_user, d->mlp_item, d->gmf_user, d->gmf_item, d->layers[0].w, d->layers[0].k, d->wfold, nu, ni,
              g->g_mlp_user, g->g_mlp_item, g->g_gmf_user, g->g_gmf_item, g->layers[0].gw, d->layers[0].k, g->layers[0].gb,
              g->g_head_w, d->counts, rows * kCountStride};
  const int64_t fwgs = ctr_ceil_div(ctr_ceil_div(nu, 16), kWaves) + ctr_ceil_div(ctr_ceil_div(ni, 16), kWaves);
  if (phases & 4) hipLaunchKernelGGL(ncfp_finish_kernel, dim3((unsigned)fwgs), dim3(kThreads), 0, st, N);
  return ctr_launch_status();
}
