// Whole narrow MLP stacks in ONE launch (forward) / ONE launch (backward).
//
// NeuralCF's tower (model/neuralcf.py:48-51: 128 -> 64 -> 32 -> 16 -> 8 -> 64 at
// BASELINE configs[1]) is 23 kFLOP per sample: as five GEMM launches it is bound by
// writing and re-reading activations and by per-launch fill/drain, not by the matrix
// cores.  Here a wave owns 32 rows (one MFMA M-tile) and carries them through every
// layer: the layer input sits in a wave-private LDS tile, all weights of the stack sit
// in LDS for the lifetime of the (persistent) workgroup, every accumulator tile goes
// bias -> activation -> LDS (next layer's operand) and -> HBM once (saved for backward).
// Waves never synchronise with each other after the weights are staged.
//
// Backward walks the layers in reverse with the same ownership: gZ = gY*act'(Y) in LDS,
// dX = gZ W (next gY, written over the tile after the operand fragments are in
// registers), dW += gZ^T X with the 32 rows as the contraction and the dW tiles living in
// accumulator registers across ALL row tiles the wave processes; one partial per
// workgroup goes to the workspace at the end and reduce.hip adds the partials up.
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32).  A contraction chunk of C <= 32 indices is
// issued as C/2 steps, lane half h taking indices base + (C/2)*h + t -- the same split
// for both operands (C % 8 == 0 so a lane's values are whole 16-byte LDS reads).
// Limits: every width <= 128 and a multiple of 8 (outputs: any n <= 128), <= 8 layers,
// everything must fit LDS; otherwise CTR_ELIMIT and the caller uses the per-layer path.
#include "ctr_common.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kMaxLayers = 8;
constexpr int kMaxDim = 128;

struct LayerDesc {
  const float* w;      // (n, k) row-major, ldw = k
  const float* b;      // (n) or null
  float* y;            // fwd: output (m, ldy); bwd: saved output (read)
  int64_t ldy;
  float* gw;           // bwd: (n, k) gradient accumulators (through the workspace)
  float* gb;
  int n, k, act;
  CtrFastDiv div_n;    // / n        (element loops over a [32][n] tile)
  CtrFastDiv div_k4;   // / (k / 4)  (dwordx4 loops over a [32][k] tile)
  int w_off;           // float offset of this layer's weights inside the LDS weight region
  int b_off;           // fwd: float offset of this layer's bias inside the LDS weight region
  int acc_off;         // bwd: first dW accumulator tile of this layer
};

struct StackDesc {
  LayerDesc l[kMaxLayers];
  int nlayers;
  int sa, sb;          // floats per row of the two wave-private LDS tiles (they swap roles per layer)
  int wfloats;         // LDS floats used by all weights (+ biases, forward)
  int ntiles;          // bwd: dW accumulator tiles per wave
  int nsum;            // bwd: sum of n over the layers (per-wave bias-gradient slots in LDS)
  int db_off[kMaxLayers];  // bwd: first bias-gradient slot of each layer
};

// Optional single-unit head on top of the stack (ctr_mlp_head_fwd): out[row] = act(x_extra[row,:p] . w[:p] +
// y_last[row,:] . w[p:] + c[0]) formed in the tile epilogue, while the last activations are still in LDS.
struct HeadDesc {
  const float* x; int64_t ldx; int p;  // p extra input columns (multiple of 8, <= 64), 16-byte aligned rows
  const float* w;                       // p + n_last weights
  const float* c;                       // one bias
  float* out; int64_t ldout; int act;
};
constexpr int kHeadMax = 192;           // p + n_last

// Backward of that head inside the stack's backward (ctr_mlp_head_bwd, pinned NeuralCF tower only: p = 64
// extra columns, 8 last activations): per row gz = gprob * act'(prob); the tower's gY tile is gz * w[p:],
// the extra columns' gradient gz * w[:p] is stored, and the p + 8 + 1 sums over the batch
// (sum gz*x_extra, sum gz*y_last, sum gz) ride along to the slab.
struct HeadBwdDesc {
  const float* gprob; int64_t ldgp;
  const float* prob; int64_t ldp;
  const float* xe; int64_t ldxe;
  const float* w;
  float* gxe; int64_t ldgxe;
  int act;
};
constexpr int kHeadBwdP = 64, kHeadBwdN = 8, kHeadBwdSums = kHeadBwdP + kHeadBwdN + 1;

// one stage of the transposed reduction: lanes r and r ^ S exchange half of their N = 2*S' values
template <int KEEP>
__device__ __forceinline__ void xpose_stage(float* pr, int bit, int xr) {
#pragma unroll
  for (int i = 0; i < KEEP; ++i) {
    const float send = bit ? pr[i] : pr[i + KEEP];
    const float keep = bit ? pr[i + KEEP] : pr[i];
    pr[i] = keep + __shfl_xor(send, xr, 64);
  }
}

// Shape policy.  DynShape: every width comes from the descriptor at run time.  A fixed shape
// pins n / k / activation of every layer at compile time for a stack that matters (the
// BASELINE NeuralCF tower): the layer loops unroll, the index divisions, tail chunks and
// range guards fold away -- the generic kernel spends ~3x its MFMA time issuing them.
struct DynShape {
  static constexpr int kWavesPerSimd = 1;
  static constexpr bool kFixed = false;
  static constexpr int kLayers = 0;
  static constexpr int N[1] = {0};
  static constexpr int K[1] = {0};
  static constexpr int ACT[1] = {0};
};
struct NcfShape {  // model/neuralcf.py:23-27 at BASELINE configs[1]: 128 -> 64 -> 32 -> 16 -> 8 -> 64
  static constexpr int kWavesPerSimd = 1;  // 224 dW accumulator registers: one wave per SIMD
  static constexpr bool kFixed = true;
  static constexpr int kLayers = 5;
  static constexpr int N[5] = {64, 32, 16, 8, 64};
  static constexpr int K[5] = {128, 64, 32, 16, 8};
  static constexpr int ACT[5] = {CTR_ACT_RELU, CTR_ACT_RELU, CTR_ACT_RELU, CTR_ACT_RELU, CTR_ACT_NONE};
};
struct NcfTowerShape {  // the same tower without its 8 -> mf_dim projection, which the model folds into the head (fold_head.hip)
  static constexpr int kWavesPerSimd = 1;  // 192 dW accumulator registers + the prefetch stage
  static constexpr bool kFixed = true;
  static constexpr int kLayers = 4;
  static constexpr int N[4] = {64, 32, 16, 8};
  static constexpr int K[4] = {128, 64, 32, 16};
  static constexpr int ACT[4] = {CTR_ACT_RELU, CTR_ACT_RELU, CTR_ACT_RELU, CTR_ACT_RELU};
};
struct DienAttShape {  // model/dien.py:13-19 at BASELINE configs[4] (E = 16) on the folded [h, t] operand: 32 -> 64 -> 32 -> 1
  static constexpr int kWavesPerSimd = 2;  // small stack: <=256 registers and <=80 KB of LDS, two workgroups per CU
  static constexpr bool kFixed = true;
  static constexpr int kLayers = 3;
  static constexpr int N[3] = {64, 32, 1};
  static constexpr int K[3] = {32, 64, 32};
  static constexpr int ACT[3] = {CTR_ACT_RELU, CTR_ACT_RELU, CTR_ACT_NONE};
};

template <class F, int... I>
__device__ __forceinline__ void static_layers(F& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

// LDS layout of a stack as compile-time functions of a fixed shape (the host's build()
// computes the same numbers at run time and refuses the instantiation if they differ).
// With strides and offsets pinned, LDS addresses are one base register + immediates; left
// as run-time values the compiler materialised ~30 row addresses per tile, hoisted them out
// of the tile loop and spilled them (a serialised scratch reload costs ~700 cycles here).
template <class S, bool BWD>
struct Layout {
  static constexpr int r8(int v) { return (v + 7) / 8 * 8; }
  static constexpr int w_off(int li) {
    int o = 0;
    for (int j = 0; j < li; ++j) o += r8(S::N[j]) * (S::K[j] + 4) + (S::N[j] + 3) / 4 * 4;
    return o;
  }
  static constexpr int b_off(int li) { return w_off(li) + r8(S::N[li]) * (S::K[li] + 4); }
  static constexpr int wfloats() { return (w_off(S::kLayers) + 3) / 4 * 4; }
  static constexpr int acc_off(int li) {
    int t = 0;
    for (int j = 0; j < li; ++j) t += ((S::N[j] + 31) / 32) * ((S::K[j] + 31) / 32);
    return t;
  }
  static constexpr int db_off(int li) {
    int t = 0;
    for (int j = 0; j < li; ++j) t += S::N[j];
    return t;
  }
  static constexpr int nsum() { return db_off(S::kLayers); }
  // widths of the two swapping tiles (see build())
  static constexpr int width(bool first) {
    int wa = 0, wb = 0;
    for (int i = 0; i < S::kLayers; ++i) {
      if (BWD) {
        const bool even = ((S::kLayers - 1 - i) & 1) == 0;
        const int n8 = r8(S::N[i]);
        if (even) {
          wa = n8 > wa ? n8 : wa;
          wb = S::K[i] > wb ? S::K[i] : wb;
        } else {
          wb = n8 > wb ? n8 : wb;
          wa = S::K[i] > wa ? S::K[i] : wa;
        }
      } else {
        if (i & 1) {
          wb = S::K[i] > wb ? S::K[i] : wb;
          wa = S::N[i] > wa ? S::N[i] : wa;
        } else {
          wa = S::K[i] > wa ? S::K[i] : wa;
          wb = S::N[i] > wb ? S::N[i] : wb;
        }
      }
    }
    return first ? wa : wb;
  }
  static constexpr int sa() { return width(true) + 4; }
  static constexpr int sb() { return width(false) + 4; }
};

template <class S, bool BWD>
__device__ __forceinline__ void pin_shape(LayerDesc& L, int li) {
  if constexpr (S::kFixed) {
    using Y = Layout<S, BWD>;
    L.n = S::N[li];
    L.k = S::K[li];
    L.act = S::ACT[li];
    L.w_off = Y::w_off(li);
    L.b_off = Y::b_off(li);
    L.acc_off = Y::acc_off(li);
  }
}

__device__ __forceinline__ int w_stride(int k) { return k + 4; }
// unguarded fragment reads of a column tile wider than the layer run < 32 floats past the
// end of a tile; weight rows past a layer land in the biases / tiles that follow.  Keep the
// former inside the allocation.
constexpr int kSlack = 32;

__device__ __forceinline__ void stage_weights(float* s_w, const StackDesc& d, bool with_bias) {
  for (int li = 0; li < d.nlayers; ++li) {
    const LayerDesc L = d.l[li];  // by value: registers, not an LDS reload after every LDS store
    if (with_bias)
      for (int i = threadIdx.x; i < L.n; i += blockDim.x) s_w[L.b_off + i] = L.b ? L.b[i] : 0.0f;
    float* dst = s_w + L.w_off;
    const int ws = w_stride(L.k);
    // rows [n, round_up(n, 8)) are staged as zeros: the contraction chunks of dX run over
    // multiples of 8 and need no per-element range check
    const int units = (L.n + 7) / 8 * 8 * (L.k / 4);  // k % 4 == 0: a float4 never crosses a row
    // four loads in flight per thread: a load-store loop pays one L2 round trip per iteration
    for (int u0 = threadIdx.x; u0 < units; u0 += 4 * blockDim.x) {
      float4 v[4];
      int off[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int u = u0 + q * blockDim.x;
        v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        off[q] = -1;
        if (u < units) {
          const int r = (int)ctr_div((uint32_t)u, L.div_k4), c = (u - r * (L.k / 4)) * 4;
          off[q] = r * ws + c;
          if (r < L.n) v[q] = *reinterpret_cast<const float4*>(L.w + (int64_t)r * L.k + c);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (off[q] >= 0) *reinterpret_cast<float4*>(dst + off[q]) = v[q];
    }
  }
}

// 16 (or fewer) contraction values of LDS row `row` for this lane: indices base + steps*h + t
template <int STEPS>
__device__ __forceinline__ void read_kc(const float* tile, int stride, int row, int base, int h, float (&f)[16]) {
  const float4* q = reinterpret_cast<const float4*>(tile + row * stride + base + STEPS * h);
#pragma unroll
  for (int v = 0; v < STEPS / 4; ++v) {
    const float4 x = q[v];
    f[4 * v + 0] = x.x; f[4 * v + 1] = x.y; f[4 * v + 2] = x.z; f[4 * v + 3] = x.w;
  }
}

// ---- tile movers: every lane issues U independent global loads before the first LDS write
// (a wave is alone on its SIMD, so a load-then-use loop would pay the full HBM latency
// once per iteration) ----

// [32][width] fp32 tile, width % 4 == 0, 16-byte aligned rows: src rows row0.. (zero past m)
template <int U, bool FIXED>
__device__ __forceinline__ void tile_load4(float* tile, int stride, const float* __restrict__ src, int64_t ld,
                                           int64_t row0, int64_t m, int width, const CtrFastDiv& div_w4, int lane) {
  const int units = 8 * width;  // dwordx4 units in the tile
  for (int i0 = lane; i0 < units; i0 += 64 * U) {
    float4 v[U];
    int off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + 64 * u;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      off[u] = -1;
      if (i < units) {
        const int rr = FIXED ? i / (width / 4) : (int)ctr_div((uint32_t)i, div_w4), c = (i - rr * (width / 4)) * 4;
        off[u] = rr * stride + c;
        if (row0 + rr < m) v[u] = *reinterpret_cast<const float4*>(src + (row0 + rr) * ld + c);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (off[u] >= 0) *reinterpret_cast<float4*>(tile + off[u]) = v[u];
  }
}

// the same tile in two halves: pre_issue puts the loads in flight into registers (<= 16
// dwordx4 per lane for width <= 128), pre_commit parks them in LDS once they are needed
// FULLM: every tile the kernel touches has its 32 rows inside the batch (m % 32 == 0): no clamps, no selects
template <bool FIXED, bool FULLM = false>
__device__ __forceinline__ void pre_issue(float4 (&pre)[16], const float* __restrict__ src, int64_t ld, int64_t row0,
                                          int64_t m, int width, const CtrFastDiv& div_w4, int lane, bool live) {
  const int units = 8 * width;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int i = lane + 64 * u;
    pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && i < units) {
      const int rr = FIXED ? i / (width / 4) : (int)ctr_div((uint32_t)i, div_w4), c = (i - rr * (width / 4)) * 4;
      if constexpr (FULLM) {
        pre[u] = *reinterpret_cast<const float4*>(src + (row0 + rr) * ld + c);
      } else {
        // clamped address + select instead of a branch per load
        const int64_t row = row0 + rr < m ? row0 + rr : m - 1;
        const float4 v = *reinterpret_cast<const float4*>(src + row * ld + c);
        if (row0 + rr < m) pre[u] = v;
      }
    }
  }
}
template <bool FIXED>
__device__ __forceinline__ void pre_commit(const float4 (&pre)[16], float* tile, int stride, int width,
                                           const CtrFastDiv& div_w4, int lane) {
  const int units = 8 * width;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int i = lane + 64 * u;
    if (i < units) {
      const int rr = FIXED ? i / (width / 4) : (int)ctr_div((uint32_t)i, div_w4), c = (i - rr * (width / 4)) * 4;
      *reinterpret_cast<float4*>(tile + rr * stride + c) = pre[u];
    }
  }
}

// same for arbitrary width / alignment, one float per lane and load
template <int U, bool FIXED>
__device__ __forceinline__ void tile_load1(float* tile, int stride, const float* __restrict__ src, int64_t ld,
                                           int64_t row0, int64_t m, int width, const CtrFastDiv& div_w, int lane) {
  const int units = 32 * width;
  for (int i0 = lane; i0 < units; i0 += 64 * U) {
    float v[U];
    int off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + 64 * u;
      v[u] = 0.0f;
      off[u] = -1;
      if (i < units) {
        const int rr = FIXED ? i / width : (int)ctr_div((uint32_t)i, div_w), c = i - rr * width;
        off[u] = rr * stride + c;
        if (row0 + rr < m) v[u] = src[(row0 + rr) * ld + c];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (off[u] >= 0) tile[off[u]] = v[u];
  }
}

// tile[r][c] *= act'(y[row0+r][c]) for a [32][width] tile (rows past m are zero already)
template <int U, bool FIXED, bool FULLM = false>
__device__ __forceinline__ void tile_mask(float* tile, int stride, const float* __restrict__ y, int64_t ld,
                                          int64_t row0, int64_t m, int width, const CtrFastDiv& div_w, int act,
                                          int lane) {
  const int units = 32 * width;
  for (int i0 = lane; i0 < units; i0 += 64 * U) {
    float v[U];
    int off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + 64 * u;
      v[u] = 0.0f;
      off[u] = -1;
      if (i < units) {
        const int rr = FIXED ? i / width : (int)ctr_div((uint32_t)i, div_w), c = i - rr * width;
        if (FULLM || row0 + rr < m) {
          off[u] = rr * stride + c;
          v[u] = y[(row0 + rr) * ld + c];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (off[u] >= 0) tile[off[u]] *= ctr_act_grad(v[u], act);
  }
}

// ------------------------------------------------------------------ forward
// one contraction chunk of STEPS*2 indices for ONE 32-column tile (row j of the weights)
// (two accumulator chains, even / odd steps: a dependent chain of this MFMA runs at ~0.4x
// its issue rate and a wave is alone on its SIMD here)
template <int STEPS>
__device__ __forceinline__ void fwd_chunk(const float* xt, int xstride, const float* wl, int wstride, int j, int base,
                                          int r, int h, floatx16& acc0, floatx16& acc1) {
  float fa[16], fb[16];
  read_kc<STEPS>(xt, xstride, r, base, h, fa);
  // lanes with j >= n read whatever follows the layer's rows in LDS: their output columns
  // are never stored, and a branch here would split every chunk into basic blocks
  read_kc<STEPS>(wl, wstride, j, base, h, fb);
#pragma unroll
  for (int t = 0; t < STEPS; t += 2) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t + 1], fb[t + 1], acc1, 0, 0, 0);
  }
}

template <class S, bool HEAD = false>
__global__ void __launch_bounds__(kThreads, S::kWavesPerSimd)
mlp_fwd_kernel(const StackDesc dk, const float* __restrict__ x, int64_t ldx, int64_t m, const HeadDesc hd) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // layer descriptors are read inside the per-tile layer loop: from LDS, not from the
  // kernarg segment (320 scalar loads + waits per wave otherwise, ~40 % of the run time)
  __shared__ StackDesc s_desc;
  for (int i = threadIdx.x; i < (int)(sizeof(StackDesc) / 4); i += blockDim.x)
    reinterpret_cast<uint32_t*>(&s_desc)[i] = reinterpret_cast<const uint32_t*>(&dk)[i];
  __syncthreads();
  const StackDesc& d = s_desc;
  float* s_w = lds;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // two wave-private tiles: layer l reads its input from one and writes its output to the
  // other, so outputs never wait for the last operand read of the same tile
  using Y = Layout<S, false>;
  const int sa = S::kFixed ? Y::sa() : d.sa, sb = S::kFixed ? Y::sb() : d.sb;
  const int nlayers = S::kFixed ? S::kLayers : d.nlayers, wfloats = S::kFixed ? Y::wfloats() : d.wfloats;
  float* ta = lds + wfloats + wave * 32 * (sa + sb);
  float* tb = ta + 32 * sa;
  const int r = lane & 31, h = lane >> 5;
  __shared__ float s_hw[HEAD ? kHeadMax : 1];
  float hc = 0.0f;
  const int nlast = S::kFixed ? S::N[S::kFixed ? S::kLayers - 1 : 0] : d.l[nlayers - 1].n;
  if constexpr (HEAD) {
    for (int i = threadIdx.x; i < hd.p + nlast; i += blockDim.x) s_hw[i] = hd.w[i];
    hc = hd.c[0];
  }
  stage_weights(s_w, d, true);
  __syncthreads();

  const int64_t tiles = (m + 31) / 32;
  const int k0 = S::kFixed ? S::K[0] : d.l[0].k;
  // layer-0 input rows of a tile as dwordx4 units held in registers (k0 <= 128: <= 16 per
  // lane): the NEXT tile's rows are in flight while the current tile goes through the stack
  float4 pre[16];
  const int units = 8 * k0;
  const CtrFastDiv div0 = d.l[0].div_k4;
  auto fetch = [&](int64_t tl) {
    const int64_t r0 = tl * 32;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = lane + 64 * u;
      pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < units && tl < tiles) {
        const int rr = S::kFixed ? i / (k0 / 4) : (int)ctr_div((uint32_t)i, div0), c = (i - rr * (k0 / 4)) * 4;
        if (r0 + rr < m) pre[u] = *reinterpret_cast<const float4*>(x + (r0 + rr) * ldx + c);
      }
    }
  };
  const int64_t tstride = (int64_t)gridDim.x * kWaves;
  fetch((int64_t)blockIdx.x * kWaves + wave);
  for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < tiles; tile += tstride) {
    const int64_t row0 = tile * 32;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = lane + 64 * u;
      if (i < units) {
        const int rr = S::kFixed ? i / (k0 / 4) : (int)ctr_div((uint32_t)i, div0), c = (i - rr * (k0 / 4)) * 4;
        *reinterpret_cast<float4*>(ta + rr * sa + c) = pre[u];
      }
    }
    fetch(tile + tstride);
    // head: this row's extra input columns, requested now and used after the last layer
    // (lane (r, h) takes the 4-float chunks 2u + h of row r)
    float4 xe[HEAD ? 8 : 1];
    if constexpr (HEAD) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = 8 * u + 4 * h;
        xe[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < hd.p && row0 + r < m) xe[u] = *reinterpret_cast<const float4*>(hd.x + (row0 + r) * hd.ldx + c);
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int li = 0; li < nlayers; ++li) {
      LayerDesc L = d.l[li];  // by value: registers, not an LDS reload after every LDS store
      pin_shape<S, false>(L, li);
      const float* wl = s_w + L.w_off;
      const int ws = w_stride(L.k);
      const int nct = (L.n + 31) / 32;
      const float* xin = (li & 1) ? tb : ta;
      float* xout = (li & 1) ? ta : tb;
      const int sin = (li & 1) ? sb : sa, sout = (li & 1) ? sa : sb;
      if (S::kFixed && !HEAD && li == nlayers - 1 && L.n == 1 && L.k % 8 == 0) {
        // a single-unit last layer of a pinned stack (DIEN's attention score, model/dien.py:13-19) is a dot product
        // per row: on the matrix cores it occupied a whole 32-column tile (16 of the 80 MFMAs of a tile) for one
        // useful column.  Lane (r, h) takes half of row r's inputs, the halves meet by shuffle.
        const int kh = L.k / 2;
        const float* xr = xin + r * sin + kh * h;
        const float* wr = wl + kh * h;
        float acc = 0.0f;
        for (int v = 0; v < kh; v += 4) {
          const float4 xv = *reinterpret_cast<const float4*>(xr + v);
          const float4 wv = *reinterpret_cast<const float4*>(wr + v);
          acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
        }
        acc += __shfl_xor(acc, 32, 64);
        if (h == 0 && row0 + r < m) L.y[(row0 + r) * L.ldy] = ctr_act(acc + s_w[L.b_off], L.act);
        __builtin_amdgcn_wave_barrier();
        continue;
      }
#pragma unroll
      for (int ct = 0; ct < nct; ++ct) {
        floatx16 a, a1;
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = a1[e] = 0.0f;
        const int j = 32 * ct + r;
        int base = 0;
#pragma unroll
        for (; base + 32 <= L.k; base += 32) fwd_chunk<16>(xin, sin, wl, ws, j, base, r, h, a, a1);
        const int rem = L.k - base;  // 0, 8, 16 or 24
        if (rem == 8) fwd_chunk<4>(xin, sin, wl, ws, j, base, r, h, a, a1);
        else if (rem == 16) fwd_chunk<8>(xin, sin, wl, ws, j, base, r, h, a, a1);
        else if (rem == 24) fwd_chunk<12>(xin, sin, wl, ws, j, base, r, h, a, a1);
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] += a1[e];
        if (j < L.n) {
          const float bias = s_w[L.b_off + j];
          float* yp = L.y + (row0 + 4 * h) * L.ldy + j;
          float* xo = xout + 4 * h * sout + j;
#pragma unroll
          for (int e = 0; e < 16; ++e) a[e] = ctr_act(a[e] + bias, L.act);
#pragma unroll
          for (int e = 0; e < 16; ++e) xo[((e & 3) + 8 * (e >> 2)) * sout] = a[e];
          if (row0 + 32 <= m) {
#pragma unroll
            for (int e = 0; e < 16; ++e) ctr_stg(yp + ((e & 3) + 8 * (e >> 2)) * L.ldy, a[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
              if (row0 + (e & 3) + 8 * (e >> 2) + 4 * h < m) ctr_stg(yp + ((e & 3) + 8 * (e >> 2)) * L.ldy, a[e]);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if constexpr (HEAD) {
      // the last layer left its output in the tile it wrote: row r, columns split between the two half-waves
      const float* fin = ((nlayers - 1) & 1) ? ta : tb;
      const int fs = ((nlayers - 1) & 1) ? sa : sb;
      float acc = 0.0f;
      for (int j = h; j < nlast; j += 2) acc = fmaf(fin[r * fs + j], s_hw[hd.p + j], acc);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = 8 * u + 4 * h;
        if (c < hd.p) {
          acc = fmaf(xe[u].x, s_hw[c], acc);
          acc = fmaf(xe[u].y, s_hw[c + 1], acc);
          acc = fmaf(xe[u].z, s_hw[c + 2], acc);
          acc = fmaf(xe[u].w, s_hw[c + 3], acc);
        }
      }
      acc += __shfl_xor(acc, 32, 64);
      if (h == 0 && row0 + r < m) hd.out[(row0 + r) * hd.ldout] = ctr_act(acc + hc, hd.act);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ------------------------------------------------------------------ forward, eight waves per workgroup
// The same stack walk with TWO waves per SIMD for a fixed shape whose first layer is the wide one (NeuralCF: 128
// inputs).  PMC of the 4-wave kernel on that tower (profiles/r02_mlp_pmc.txt): a wave lives 89 k cycles, its
// matrix instructions account for 24 k and nothing overlaps them -- it is alone on its SIMD, because the 32 x 132
// input tile of layer 0 makes a wave's LDS strip 25 KB.  Here layer 0 takes its A operand straight from global
// memory in fragment order (row r, columns 32c + 16h .. +15: four dwordx4 per contraction chunk, the registers
// that held the prefetched tile anyway), so the strip only holds the narrow activations (13 KB) and eight waves
// plus the weights fit the CU.
constexpr int kWavesDirect = 8;

template <class S>
struct DirectLayout {
  static constexpr int width(bool a) {   // tile A holds outputs of odd layers, tile B of even layers
    int w = 8;
    for (int i = 0; i < S::kLayers; ++i)
      if (((i & 1) != 0) == a) w = S::N[i] > w ? S::N[i] : w;
    return w;
  }
  static constexpr int sa() { return width(true) + 4; }
  static constexpr int sb() { return width(false) + 4; }
  static constexpr size_t lds_bytes() {
    return sizeof(float) * (size_t)(Layout<S, false>::wfloats() + kWavesDirect * 32 * (sa() + sb()) + kSlack);
  }
};

template <class S, bool HEAD>
__global__ void __launch_bounds__(64 * kWavesDirect)
mlp_fwd_direct_kernel(const StackDesc dk, const float* __restrict__ x, int64_t ldx, int64_t m, const HeadDesc hd) {
  static_assert(S::kFixed && S::K[0] % 32 == 0 && S::K[0] <= 128, "layer 0: whole 32-index chunks in registers");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ StackDesc s_desc;
  for (int i = threadIdx.x; i < (int)(sizeof(StackDesc) / 4); i += blockDim.x)
    reinterpret_cast<uint32_t*>(&s_desc)[i] = reinterpret_cast<const uint32_t*>(&dk)[i];
  using Y = Layout<S, false>;
  using D = DirectLayout<S>;
  constexpr int sa = D::sa(), sb = D::sb(), nlayers = S::kLayers, wfloats = Y::wfloats();
  constexpr int kChunks0 = S::K[0] / 32;
  float* s_w = lds;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* ta = lds + wfloats + wave * 32 * (sa + sb);
  float* tb = ta + 32 * sa;
  const int r = lane & 31, h = lane >> 5;
  __shared__ float s_hw[HEAD ? kHeadMax : 1];
  constexpr int nlast = S::N[S::kLayers - 1];
  const int64_t tiles = (m + 31) / 32;
  const int64_t tstride = (int64_t)gridDim.x * kWavesDirect;
  // the first tile's operand rows are requested before the weights are staged
  float4 pre[4 * kChunks0];
  auto fetch = [&](int64_t tl) {
    const int64_t row = tl * 32 + r;
    const bool live = tl < tiles && row < m;
    const float* src = x + (live ? row : 0) * ldx + 16 * h;
#pragma unroll
    for (int c = 0; c < kChunks0; ++c)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        pre[4 * c + v] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) pre[4 * c + v] = *reinterpret_cast<const float4*>(src + 32 * c + 4 * v);
      }
  };
  fetch((int64_t)blockIdx.x * kWavesDirect + wave);
  __syncthreads();
  const StackDesc& d = s_desc;
  float hc = 0.0f;
  if constexpr (HEAD) {
    for (int i = threadIdx.x; i < hd.p + nlast; i += blockDim.x) s_hw[i] = hd.w[i];
    hc = hd.c[0];
  }
  stage_weights(s_w, d, true);
  __syncthreads();

  for (int64_t tile = (int64_t)blockIdx.x * kWavesDirect + wave; tile < tiles; tile += tstride) {
    const int64_t row0 = tile * 32;
    float4 xe[HEAD ? 8 : 1];
    if constexpr (HEAD) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = 8 * u + 4 * h;
        xe[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < hd.p && row0 + r < m) xe[u] = *reinterpret_cast<const float4*>(hd.x + (row0 + r) * hd.ldx + c);
      }
    }
    auto layer = [&](auto liv) __attribute__((always_inline)) {
      constexpr int li = decltype(liv)::value;
      LayerDesc L = d.l[li];
      pin_shape<S, false>(L, li);
      const float* wl = s_w + L.w_off;
      constexpr int ws = S::K[li] + 4;
      constexpr int nct = (S::N[li] + 31) / 32;
      const float* xin = (li & 1) ? tb : ta;     // layer li > 0 reads what layer li - 1 wrote
      float* xout = (li & 1) ? ta : tb;
      constexpr int sin = (li & 1) ? sb : sa, sout = (li & 1) ? sa : sb;
#pragma unroll
      for (int ct = 0; ct < nct; ++ct) {
        floatx16 a, a1;
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = a1[e] = 0.0f;
        const int j = 32 * ct + r;
        if constexpr (li == 0) {
#pragma unroll
          for (int c = 0; c < kChunks0; ++c) {
            float fb[16];
            read_kc<16>(wl, ws, j, 32 * c, h, fb);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const float4 q = pre[4 * c + v];
              a = __builtin_amdgcn_mfma_f32_32x32x2f32(q.x, fb[4 * v + 0], a, 0, 0, 0);
              a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q.y, fb[4 * v + 1], a1, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_32x32x2f32(q.z, fb[4 * v + 2], a, 0, 0, 0);
              a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q.w, fb[4 * v + 3], a1, 0, 0, 0);
            }
          }
        } else {
          constexpr int k = S::K[li];
          int base = 0;
#pragma unroll
          for (; base + 32 <= k; base += 32) fwd_chunk<16>(xin, sin, wl, ws, j, base, r, h, a, a1);
          constexpr int rem = k % 32;  // 0, 8, 16 or 24
          if constexpr (rem == 8) fwd_chunk<4>(xin, sin, wl, ws, j, base, r, h, a, a1);
          else if constexpr (rem == 16) fwd_chunk<8>(xin, sin, wl, ws, j, base, r, h, a, a1);
          else if constexpr (rem == 24) fwd_chunk<12>(xin, sin, wl, ws, j, base, r, h, a, a1);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] += a1[e];
        if (j < S::N[li]) {
          const float bias = s_w[L.b_off + j];
          float* yp = L.y + (row0 + 4 * h) * L.ldy + j;
          float* xo = xout + 4 * h * sout + j;
#pragma unroll
          for (int e = 0; e < 16; ++e) a[e] = ctr_act(a[e] + bias, S::ACT[li]);
#pragma unroll
          for (int e = 0; e < 16; ++e) xo[((e & 3) + 8 * (e >> 2)) * sout] = a[e];
          if (row0 + 32 <= m) {
#pragma unroll
            for (int e = 0; e < 16; ++e) ctr_stg(yp + ((e & 3) + 8 * (e >> 2)) * L.ldy, a[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
              if (row0 + (e & 3) + 8 * (e >> 2) + 4 * h < m) ctr_stg(yp + ((e & 3) + 8 * (e >> 2)) * L.ldy, a[e]);
          }
        }
      }
      if constexpr (li == 0) fetch(tile + tstride);  // layer 0 is through with the registers
      __builtin_amdgcn_wave_barrier();
    };
    static_layers(layer, std::make_integer_sequence<int, S::kLayers>{});
    if constexpr (HEAD) {
      const float* fin = ((nlayers - 1) & 1) ? ta : tb;
      constexpr int fs = ((nlayers - 1) & 1) ? sa : sb;
      float acc = 0.0f;
#pragma unroll
      for (int j = h; j < nlast; j += 2) acc = fmaf(fin[r * fs + j], s_hw[hd.p + j], acc);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = 8 * u + 4 * h;
        if (c < hd.p) {
          acc = fmaf(xe[u].x, s_hw[c], acc);
          acc = fmaf(xe[u].y, s_hw[c + 1], acc);
          acc = fmaf(xe[u].z, s_hw[c + 2], acc);
          acc = fmaf(xe[u].w, s_hw[c + 3], acc);
        }
      }
      acc += __shfl_xor(acc, 32, 64);
      if (h == 0 && row0 + r < m) hd.out[(row0 + r) * hd.ldout] = ctr_act(acc + hc, hd.act);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ------------------------------------------------------------------ backward
// KS-mode fragment: values tile[(base + STEPS*h + t) * stride + col], t < STEPS
template <int STEPS>
__device__ __forceinline__ void read_ks(const float* tile, int stride, int col, int base, int h, float (&f)[16]) {
#pragma unroll
  for (int t = 0; t < STEPS; ++t) f[t] = tile[(base + STEPS * h + t) * stride + col];
}

// dX chunk for ONE 32-column tile: contraction over n, A = gZ rows (KC), B = W[n][col] (KS).
// n need not be a multiple of 8: the gZ tile is zero-padded to the chunk width and weight
// rows >= n read as zero.
template <int STEPS>
__device__ __forceinline__ void dx_chunk(const float* gt, int gstride, const float* wl, int wstride, int col, int base,
                                         int r, int h, floatx16& acc0, floatx16& acc1) {
  float fa[16], fb[16];
  read_kc<STEPS>(gt, gstride, r, base, h, fa);
  // one base address + immediate offsets; lanes with col >= k feed columns nobody stores
  const float* wp = wl + (base + STEPS * h) * wstride + col;
#pragma unroll
  for (int t = 0; t < STEPS; ++t) fb[t] = wp[t * wstride];
#pragma unroll
  for (int t = 0; t < STEPS; t += 2) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t + 1], fb[t + 1], acc1, 0, 0, 0);
  }
}

// the same with ONE accumulator chain: a dependent chain of v_mfma_f32_32x32x2_f32 issues at its full rate (64-cycle
// issue, 64-cycle dependent latency), and the second chain cost 16 zero moves + 16 accumulator reads + 16 adds per
// column tile in a kernel that is bound by instruction issue (profiles/r02_mlp_pmc.txt)
template <int STEPS>
__device__ __forceinline__ void dx_chunk1(const float* gt, int gstride, const float* wl, int wstride, int col, int base,
                                          int r, int h, floatx16& acc) {
  float fa[16], fb[16];
  read_kc<STEPS>(gt, gstride, r, base, h, fa);
  const float* wp = wl + (base + STEPS * h) * wstride + col;
#pragma unroll
  for (int t = 0; t < STEPS; ++t) fb[t] = wp[t * wstride];
#pragma unroll
  for (int t = 0; t < STEPS; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc, 0, 0, 0);
}

template <class S, int MAXT, bool HEADB = false, bool FULLM = false>
__global__ void __launch_bounds__(kThreads, S::kWavesPerSimd)
mlp_bwd_kernel(const StackDesc dk, const float* __restrict__ x, int64_t ldx, int64_t m, const float* __restrict__ gy,
               int64_t ldgy, float* __restrict__ gx, int64_t ldgx, float* __restrict__ ws, int64_t slab,
               const HeadBwdDesc hb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ StackDesc s_desc;  // see mlp_fwd_kernel
#ifdef CTR_MLP_TIMING
  uint64_t stamps[16];
  int nstamp = 0;
#define CTR_STAMP() do { if (nstamp < 16) stamps[nstamp++] = __builtin_readcyclecounter(); } while (0)
#else
#define CTR_STAMP() do {} while (0)
#endif
  CTR_STAMP();
  for (int i = threadIdx.x; i < (int)(sizeof(StackDesc) / 4); i += blockDim.x)
    reinterpret_cast<uint32_t*>(&s_desc)[i] = reinterpret_cast<const uint32_t*>(&dk)[i];
  __syncthreads();
  const StackDesc& d = s_desc;
  float* s_w = lds;
  const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // two wave-private tiles P (stride sa) and Q (stride sb).  Walking the layers from the
  // last one, the gradient tile and the layer-input tile swap every layer: dX_l is written
  // over X_l (dW_l has consumed it) and is the gY of layer l-1, whose input then goes
  // where gZ_l was.
  using Y = Layout<S, true>;
  const int sa = S::kFixed ? Y::sa() : d.sa, sb = S::kFixed ? Y::sb() : d.sb;
  const int wfloats = S::kFixed ? Y::wfloats() : d.wfloats, nsum = S::kFixed ? Y::nsum() : d.nsum;
  float* tp = lds + wfloats + wave * 32 * (sa + sb);
  float* tq = tp + 32 * sa;
  stage_weights(s_w, d, false);
  __syncthreads();
  CTR_STAMP();

  floatx16 dw[MAXT];  // dW accumulator tiles of every layer, alive across all row tiles
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) dw[t][e] = 0.0f;
  // bias-gradient partials: nsum floats per wave in LDS (lane j owns column j of each layer)
  float* s_db = lds + wfloats + kWaves * 32 * (sa + sb) + wave * nsum;
  for (int i = lane0; i < nsum; i += 64) s_db[i] = 0.0f;
  // fused head backward: its weights, and this lane's share of the batch sums
  __shared__ __attribute__((aligned(16))) float s_hwb[HEADB ? kHeadBwdP + kHeadBwdN : 4];
  float hx_sum[2] = {0.0f, 0.0f}, hy_sum = 0.0f, hc_sum = 0.0f;
  if constexpr (HEADB) {
    for (int i = threadIdx.x; i < kHeadBwdP + kHeadBwdN; i += blockDim.x) s_hwb[i] = hb.w[i];
    __syncthreads();
  }

  const int64_t tiles = (m + 31) / 32;
  const int nlayers = S::kFixed ? S::kLayers : d.nlayers;
  const int last = nlayers - 1;
  const CtrFastDiv div_last = d.l[last].div_n;
  const int nl = S::kFixed ? S::N[S::kFixed ? S::kLayers - 1 : 0] : d.l[last].n;
  constexpr bool kPre = S::kFixed;
  const int64_t tstride = (int64_t)gridDim.x * kWaves;
  float4 pre[16];   // kPre: the next layer-input tile, in flight
  float4 gpre[16];  // kPre: the next tile's gY
  const bool gvec = !HEADB && kPre && nl % 4 == 0 && ldgy % 4 == 0 && (reinterpret_cast<uintptr_t>(gy) & 15) == 0;
  const float* xlast = last > 0 ? d.l[last > 0 ? last - 1 : 0].y : x;
  const int64_t ldxlast = last > 0 ? d.l[last > 0 ? last - 1 : 0].ldy : ldx;
  if constexpr (kPre) {
    const int64_t t0 = (int64_t)blockIdx.x * kWaves + wave;
    pre_issue<true, FULLM>(pre, xlast, ldxlast, t0 * 32, m, S::K[S::kFixed ? S::kLayers - 1 : 0], d.l[last].div_k4, lane0,
                    t0 < tiles);
    if (gvec) pre_issue<true, FULLM>(gpre, gy, ldgy, t0 * 32, m, nl, div_last, lane0, t0 < tiles);
  }
  // fused head backward: the tile's head inputs (gprob, prob, the last activations, the 64 extra columns of this
  // lane's row) are requested a tile ahead -- loaded at the top of the tile they were an exposed memory latency per
  // tile in a kernel whose waves see two tiles
  float hp_g = 0.0f, hp_p = 0.0f;
  float4 hp_y = make_float4(0.f, 0.f, 0.f, 0.f), hp_x[8];
  auto head_fetch = [&](int64_t tl) {
    const int64_t row = tl * 32 + (lane0 & 31);
    const bool ok = tl < tiles && (FULLM || row < m);
    const int hh = lane0 >> 5;
    hp_g = hp_p = 0.0f;
    hp_y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) hp_x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
      hp_g = hb.gprob[row * hb.ldgp];
      hp_p = hb.prob[row * hb.ldp];
      hp_y = *reinterpret_cast<const float4*>(d.l[last].y + row * d.l[last].ldy + 4 * hh);
#pragma unroll
      for (int i = 0; i < 8; ++i)   // half = i >> 2: columns 32 hh + 16 half + 4 (i & 3)
        hp_x[i] = *reinterpret_cast<const float4*>(hb.xe + row * hb.ldxe + 32 * hh + 16 * (i >> 2) + 4 * (i & 3));
    }
  };
  if constexpr (HEADB) head_fetch((int64_t)blockIdx.x * kWaves + wave);
  for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < tiles; tile += tstride) {
    const int64_t row0 = tile * 32;
    // fixed shape: every lane-derived LDS / global offset of the unrolled stack is loop
    // invariant, and hoisting those few hundred values out of the tile loop spills them.
    // Re-deriving them per tile costs a shift and an add each.
    const bool full = FULLM || row0 + 32 <= m;
    int lane = lane0;
    if constexpr (S::kFixed) asm volatile("" : "+v"(lane));
    const int r = lane & 31, h = lane >> 5;
    // gradient of the last layer's output -> P (requested during the previous tile's
    // layer 0 when it can be fetched as dwordx4)
    if constexpr (HEADB) {
      // lane (r, h): row r of the tile.  gz, the tower's gY tile (4 of its 8 columns per half-wave), the
      // gradient of the extra columns (32 per half-wave), and the batch sums by a transposed reduction:
      // lane r ends up with column (r & 15) of each 16-column group, summed over the 32 rows
      const int64_t row = row0 + r;
      const bool ok = FULLM || row < m;
      const float gz = hp_g * ctr_act_grad(hp_p, hb.act);   // rows past m were fetched as zeros
      const float4 yv = hp_y;
      {
        // the tower's gY tile with the last layer's activation derivative folded in (its outputs are right here):
        // no separate masking pass over Y
        const int act_l = S::kFixed ? S::ACT[S::kFixed ? S::kLayers - 1 : 0] : d.l[last].act;
        const float yq[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          tp[r * sa + 4 * h + q] = gz * s_hwb[kHeadBwdP + 4 * h + q] * ctr_act_grad(yq[q], act_l);
      }
      if (h == 0) hc_sum += gz;
      {
        float pr[4] = {gz * yv.x, gz * yv.y, gz * yv.z, gz * yv.w};
        xpose_stage<2>(pr, r & 2, 2);
        xpose_stage<1>(pr, r & 1, 1);
        float t = pr[0];
        t += __shfl_xor(t, 4, 64);
        t += __shfl_xor(t, 8, 64);
        t += __shfl_xor(t, 16, 64);
        hy_sum += t;  // column 4h + (r & 3)
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int c0 = 32 * h + 16 * half;
        float pr[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 xv = hp_x[4 * half + i];
          const float4 wv = *reinterpret_cast<const float4*>(s_hwb + c0 + 4 * i);
          if (ok)
            *reinterpret_cast<float4*>(hb.gxe + row * hb.ldgxe + c0 + 4 * i) =
                make_float4(gz * wv.x, gz * wv.y, gz * wv.z, gz * wv.w);
          pr[4 * i + 0] = gz * xv.x; pr[4 * i + 1] = gz * xv.y; pr[4 * i + 2] = gz * xv.z; pr[4 * i + 3] = gz * xv.w;
        }
        xpose_stage<8>(pr, r & 8, 8);
        xpose_stage<4>(pr, r & 4, 4);
        xpose_stage<2>(pr, r & 2, 2);
        xpose_stage<1>(pr, r & 1, 1);
        hx_sum[half] += pr[0] + __shfl_xor(pr[0], 16, 64);  // column c0 + (r & 15)
      }
      head_fetch(tile + tstride);   // everything of this tile is consumed: the registers take the next tile's
    } else if (kPre && gvec) {
      pre_commit<true>(gpre, tp, sa, nl, div_last, lane);
    } else {
      tile_load1<8, S::kFixed>(tp, sa, gy, ldgy, row0, m, nl, div_last, lane);
    }
    __builtin_amdgcn_wave_barrier();
    {
      // gZ = gY * act'(Y) of the last layer, in place (rows past m stay zero)
      const int act_last = S::kFixed ? S::ACT[S::kFixed ? S::kLayers - 1 : 0] : d.l[last].act;
      if (!HEADB && act_last != CTR_ACT_NONE) {   // (the fused head wrote the tile already masked)
        tile_mask<8, S::kFixed, FULLM>(tp, sa, d.l[last].y, d.l[last].ldy, row0, m, nl, div_last, act_last, lane);
        __builtin_amdgcn_wave_barrier();
      }
    }
    // one layer of the backward walk; lqv is an int (DynShape) or an integral_constant
    // (fixed shape: one instantiation per layer, everything shape-dependent folds)
    auto layer = [&](auto lqv) __attribute__((always_inline)) {
      const int li = last - (int)lqv;
      {
        LayerDesc L = d.l[li];  // by value: registers, not an LDS reload after every LDS store
        pin_shape<S, true>(L, li);
        const int dboff = S::kFixed ? Y::db_off(S::kFixed ? li : 0) : d.db_off[li];
        const bool even = ((last - li) & 1) == 0;
        float* gt = even ? tp : tq;      // gY -> gZ of this layer
        float* xt = even ? tq : tp;      // X_l, then dX_l
        const int gs = even ? sa : sb, xs = even ? sb : sa;
        // gt holds gZ_l already: the last layer's gY was masked on load, every other one
        // by the dX write-back of the layer above
        {
          // zero the columns [n, round_up(n, 8)) the last dX chunk will read
          const int npad = (L.n + 7) / 8 * 8 - L.n;
          for (int i = lane; i < 32 * npad; i += 64) {
            const int rr = i / npad, c = i - rr * npad;
            gt[rr * gs + L.n + c] = 0.0f;
          }
        }
        // layer input X_l (layer 0: the stack input, else the saved output of layer l-1)
        const float* src = x;
        int64_t ld_in = ldx;
        if (li > 0) {
          src = d.l[li > 0 ? li - 1 : 0].y;
          ld_in = d.l[li > 0 ? li - 1 : 0].ldy;
        }
        if constexpr (kPre) {
          // X_l was requested one layer (or one tile) ago; park it, then request the next
          // operand so its HBM latency runs under this layer's MFMAs
          pre_commit<true>(pre, xt, xs, L.k, L.div_k4, lane);
          if (li > 0) {
            const int lp = li > 0 ? li - 1 : 0;
            const float* nsrc = lp > 0 ? d.l[lp > 0 ? lp - 1 : 0].y : x;
            const int64_t nld = lp > 0 ? d.l[lp > 0 ? lp - 1 : 0].ldy : ldx;
            pre_issue<true, FULLM>(pre, nsrc, nld, row0, m, S::K[S::kFixed ? lp : 0], d.l[lp].div_k4, lane, true);
          } else {
            const int64_t nt = tile + tstride;
            pre_issue<true, FULLM>(pre, xlast, ldxlast, nt * 32, m, S::K[S::kFixed ? S::kLayers - 1 : 0], d.l[last].div_k4, lane,
                            nt < tiles);
            if (gvec) pre_issue<true, FULLM>(gpre, gy, ldgy, nt * 32, m, nl, div_last, lane, nt < tiles);
          }
        } else {
          tile_load4<4, S::kFixed>(xt, xs, src, ld_in, row0, m, L.k, L.div_k4, lane);
        }
        __builtin_amdgcn_wave_barrier();
        // (bias gradient: taken from the gZ fragments of the dW products below -- a column's 32 values are the 16
        // of each half-wave's fragment -- instead of 32 more LDS reads per column)
        CTR_STAMP();
        // dW_l[n][k] += sum_rows gZ[row][n] X[row][k]: contraction = the 32 rows, 16 steps.
        // The accumulator index must be a compile-time constant (a run-time index would
        // send the tiles to scratch), so walk every slot and take this layer's ones.
        {
          const int nrt = (L.n + 31) / 32, nkt = (L.k + 31) / 32;
#pragma unroll
          for (int s2 = 0; s2 < MAXT; s2 += 2) {
            // two slots at a time, their MFMAs interleaved: independent accumulators back
            // to back instead of one 16-deep dependent chain
            const int relA = s2 - L.acc_off, relB = relA + 1;
            const bool okA = relA >= 0 && relA < nrt * nkt, okB = relB >= 0 && relB < nrt * nkt;
            if (okA || okB) {
              float faA[16], fbA[16], faB[16], fbB[16];
              const int itA = okA ? relA / nkt : 0, jtA = okA ? relA - itA * nkt : 0;
              const int itB = okB ? relB / nkt : 0, jtB = okB ? relB - itB * nkt : 0;
              // columns past n / k read their neighbours in LDS and land in dW elements
              // that the flush drops; only a slot of another layer must add nothing
              if (okA) {
                read_ks<16>(gt, gs, 32 * itA + r, 0, h, faA);
                read_ks<16>(xt, xs, 32 * jtA + r, 0, h, fbA);
              } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) faA[t] = fbA[t] = 0.0f;
              }
              if (okB) {
                if (okA && itB == itA) {   // the pair shares its row of tiles: one gZ fragment
#pragma unroll
                  for (int t = 0; t < 16; ++t) faB[t] = faA[t];
                } else {
                  read_ks<16>(gt, gs, 32 * itB + r, 0, h, faB);
                }
                read_ks<16>(xt, xs, 32 * jtB + r, 0, h, fbB);
              } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) faB[t] = fbB[t] = 0.0f;
              }
              // bias gradient of the columns 32*it .. +31, once per row of tiles (at its first tile)
              if ((okA && jtA == 0) || (okB && jtB == 0)) {
                const bool fromA = okA && jtA == 0;
                const int itD = fromA ? itA : itB;
                float t = 0.0f;
#pragma unroll
                for (int q = 0; q < 16; ++q) t += fromA ? faA[q] : faB[q];
                t += __shfl_xor(t, 32, 64);
                if (h == 0 && 32 * itD + r < L.n) s_db[dboff + 32 * itD + r] += t;
                if (okA && okB && jtA == 0 && jtB == 0) {   // one-tile-wide layer: both slots start a row of tiles
                  float u = 0.0f;
#pragma unroll
                  for (int q = 0; q < 16; ++q) u += faB[q];
                  u += __shfl_xor(u, 32, 64);
                  if (h == 0 && 32 * itB + r < L.n) s_db[dboff + 32 * itB + r] += u;
                }
              }
              // a slot of another layer gets no MFMA at all (fixed shapes: okA / okB are constants; multiplying its
              // zero fragments cost the pinned tower 32 of 408 matrix instructions per tile)
              if (okA && okB) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                  dw[s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(faA[t], fbA[t], dw[s2], 0, 0, 0);
                  dw[s2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(faB[t], fbB[t], dw[s2 + 1], 0, 0, 0);
                }
              } else if (okA) {
#pragma unroll
                for (int t = 0; t < 16; ++t) dw[s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(faA[t], fbA[t], dw[s2], 0, 0, 0);
              } else {
#pragma unroll
                for (int t = 0; t < 16; ++t)
                  dw[s2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(faB[t], fbB[t], dw[s2 + 1], 0, 0, 0);
              }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
        CTR_STAMP();
        // dX_l = gZ W_l (32 x k): over X_l in LDS (the next layer's gY) or, for layer 0, to HBM
        {
          const float* wl = s_w + L.w_off;
          const int wsd = w_stride(L.k);
          const int nkt = (L.k + 31) / 32;
          int act_prev = CTR_ACT_NONE;
          if (li > 0) act_prev = S::kFixed ? S::ACT[S::kFixed && li > 0 ? li - 1 : 0] : d.l[li > 0 ? li - 1 : 0].act;
#pragma unroll
          for (int ct = 0; ct < nkt; ++ct) {
            floatx16 a;
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] = 0.0f;
            const int col = 32 * ct + r;
            int base = 0;
#pragma unroll
            for (; base + 32 <= L.n; base += 32) dx_chunk1<16>(gt, gs, wl, wsd, col, base, r, h, a);
            const int rem = L.n - base;  // 0..31, the tile is zero-padded to a multiple of 8
            if (rem > 24) dx_chunk1<16>(gt, gs, wl, wsd, col, base, r, h, a);
            else if (rem > 16) dx_chunk1<12>(gt, gs, wl, wsd, col, base, r, h, a);
            else if (rem > 8) dx_chunk1<8>(gt, gs, wl, wsd, col, base, r, h, a);
            else if (rem > 0) dx_chunk1<4>(gt, gs, wl, wsd, col, base, r, h, a);
            if (col < L.k) {
              if (li == 0) {
                if (gx) {
                  float* gp = gx + (row0 + 4 * h) * ldgx + col;
                  if (full) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) ctr_stg(gp + ((e & 3) + 8 * (e >> 2)) * ldgx, a[e]);
                  } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                      const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                      if (row0 + row < m) ctr_stg(gp + ((e & 3) + 8 * (e >> 2)) * ldgx, a[e]);
                    }
                  }
                }
              } else {
                // X_l = Y_{l-1} sits where dX_l goes: fold act'_{l-1}(Y_{l-1}) in, so the
                // tile leaves as gZ_{l-1} and layer l-1 needs no pass over Y from HBM.
                // All reads first: the compiler cannot move an LDS read over an LDS write.
                float* q = xt + 4 * h * xs + col;
                if (act_prev != CTR_ACT_NONE) {
                  float yv[16];
#pragma unroll
                  for (int e = 0; e < 16; ++e) yv[e] = q[((e & 3) + 8 * (e >> 2)) * xs];
#pragma unroll
                  for (int e = 0; e < 16; ++e) a[e] *= ctr_act_grad(yv[e], act_prev);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) q[((e & 3) + 8 * (e >> 2)) * xs] = a[e];
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    };
    if constexpr (S::kFixed) {
      static_layers(layer, std::make_integer_sequence<int, S::kLayers>{});
    } else {
      for (int lq = 0; lq < nlayers; ++lq) layer(lq);
    }
    CTR_STAMP();
  }
  CTR_STAMP();

  // workgroup partial of dW / db -> one slab in the workspace.  No LDS atomics: ds_add_f32
  // from 4 waves ran at ~200 cycles per wave-instruction here (100K cycles for the 128x64
  // layer).  Instead every wave stores its tiles into a copy of its own (weights and
  // activation tiles are dead, so up to 4 copies of a layer fit), in phases when fewer
  // copies fit, and the copies are summed in a fixed order on the way to the workspace.
  const int lane = lane0, r = lane0 & 31, h = lane0 >> 5;
  __syncthreads();
  const int avail = wfloats + kWaves * 32 * (sa + sb);
  float* s_red = lds;
  const float* s_dbv = lds + avail;  // the waves' bias partials, kept beyond the tiles
  // layout of a slab: for each layer  n*k weights then n biases, in layer order
  int off = 0;
  auto flush = [&](auto liv) __attribute__((always_inline)) {
    const int li = (int)liv;
    LayerDesc L = d.l[li];  // by value: registers, not an LDS reload after every LDS store
    pin_shape<S, true>(L, li);
    const int nkt = (L.k + 31) / 32, nrt = (L.n + 31) / 32;
    const int cnt = L.n * L.k + L.n;
    const int copies = 4 * cnt <= avail ? 4 : (2 * cnt <= avail ? 2 : 1);
    float* mine = s_red + (wave & (copies - 1)) * cnt;
    const int dboff = wave * nsum + (S::kFixed ? Y::db_off(S::kFixed ? li : 0) : d.db_off[li]);
    for (int p = 0; p < kWaves / copies; ++p) {
      if (wave / copies == p) {
        // two code paths, not a select per element: the first phase is plain stores
        // straight from the accumulator registers
#pragma unroll
        for (int s2 = 0; s2 < MAXT; ++s2) {
          const int rel = s2 - L.acc_off;
          if (rel >= 0 && rel < nrt * nkt) {
            const int it = rel / nkt, jt = rel - it * nkt;
            const int kcol = 32 * jt + r;
            float* q = mine + (32 * it + 4 * h) * L.k + kcol;
            if (kcol < L.k) {
              if (p == 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                  const int rr = (e & 3) + 8 * (e >> 2);
                  if (32 * it + 4 * h + rr < L.n) q[rr * L.k] = dw[s2][e];
                }
              } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                  const int rr = (e & 3) + 8 * (e >> 2);
                  if (32 * it + 4 * h + rr < L.n) q[rr * L.k] += dw[s2][e];
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);  // one slot at a time: no pile-up of accumulator copies
          }
        }
        for (int j = lane; j < L.n; j += 64) {
          float* q = mine + L.n * L.k + j;
          *q = p == 0 ? s_dbv[dboff + j] : *q + s_dbv[dboff + j];
        }
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
      float v = s_red[i];
      for (int c = 1; c < copies; ++c) v += s_red[c * cnt + i];
      ws[(int64_t)blockIdx.x * slab + off + i] = v;
    }
    __syncthreads();
    off += cnt;
  };
  if constexpr (S::kFixed) {
    static_layers(flush, std::make_integer_sequence<int, S::kLayers>{});
  } else {
    for (int li = 0; li < nlayers; ++li) flush(li);
  }
  if constexpr (HEADB) {
    // the head's sums: every wave parks its share, the workgroup adds the four in wave order
    float* s_h = s_red;  // [kWaves][80]; the flush above ended with a barrier
    if (r < 16) {
      s_h[wave * 80 + 32 * h + r] = hx_sum[0];
      s_h[wave * 80 + 32 * h + 16 + r] = hx_sum[1];
    }
    if (r < 4) s_h[wave * 80 + kHeadBwdP + 4 * h + r] = hy_sum;
    const float c = ctr_wave_sum(hc_sum);
    if (lane == 0) s_h[wave * 80 + kHeadBwdP + kHeadBwdN] = c;
    __syncthreads();
    for (int i = threadIdx.x; i < kHeadBwdSums; i += blockDim.x) {
      float v = s_h[i];
      for (int w2 = 1; w2 < kWaves; ++w2) v += s_h[w2 * 80 + i];
      ws[(int64_t)blockIdx.x * slab + off + i] = v;
    }
  }
  CTR_STAMP();
#ifdef CTR_MLP_TIMING
  if (threadIdx.x == 0)
    for (int i = 0; i < 16; ++i)
      ws[(int64_t)gridDim.x * slab + blockIdx.x * 16 + i] = i < nstamp ? (float)(stamps[i] - stamps[0]) : -1.0f;
#endif
}

// ------------------------------------------------------------------ host
struct Built {
  StackDesc d;
  size_t lds_bytes;
  int64_t slab;
};

int build(const ctr_mlp_layer_t* layers, int nlayers, bool backward, Built* out) {
  CTR_REQUIRE(layers && nlayers >= 1 && nlayers <= kMaxLayers, CTR_ELIMIT);
  StackDesc& d = out->d;
  d.nlayers = nlayers;
  int woff = 0, maxk = 0, maxn = 0, tiles = 0;
  int64_t slab = 0;
  for (int i = 0; i < nlayers; ++i) {
    const ctr_mlp_layer_t& s = layers[i];
    CTR_REQUIRE(s.w && s.n >= 1 && s.k >= 8, CTR_EINVAL);
    CTR_REQUIRE(s.n <= kMaxDim && s.k <= kMaxDim && s.k % 8 == 0, CTR_ELIMIT);
    CTR_REQUIRE(i == 0 || (s.k == layers[i - 1].n), CTR_EINVAL);
    CTR_REQUIRE(s.act >= CTR_ACT_NONE && s.act <= CTR_ACT_SIGMOID, CTR_EINVAL);
    CTR_REQUIRE(s.y && s.ldy >= s.n && ctr_aligned16(s.w), CTR_EINVAL);
    // intermediate outputs are re-read with dwordx4 as the next layer's input
    CTR_REQUIRE(i == nlayers - 1 || (ctr_aligned16(s.y) && s.ldy % 4 == 0), CTR_EALIGN);
    LayerDesc& L = d.l[i];
    L.w = s.w; L.b = s.b; L.y = s.y; L.ldy = s.ldy; L.gw = s.gw; L.gb = s.gb;
    L.n = s.n; L.k = s.k; L.act = s.act;
    L.div_n = ctr_fastdiv((uint32_t)s.n);
    L.div_k4 = ctr_fastdiv((uint32_t)(s.k / 4));
    L.w_off = woff;
    L.acc_off = tiles;
    woff += (s.n + 7) / 8 * 8 * (s.k + 4);
    L.b_off = woff;
    woff += (s.n + 3) / 4 * 4;
    tiles += ((s.n + 31) / 32) * ((s.k + 31) / 32);
    maxk = s.k > maxk ? s.k : maxk;
    maxn = s.n > maxn ? s.n : maxn;
    slab += (int64_t)s.n * s.k + s.n;
    if (backward) CTR_REQUIRE(s.gw && s.gb, CTR_EINVAL);
  }
  d.wfloats = (woff + 3) / 4 * 4;
  d.ntiles = tiles;
  (void)maxk;
  (void)maxn;
  // widths the two swapping tiles must hold
  int wa = 0, wb = 0;
  if (backward) {
    const int last = nlayers - 1;
    for (int i = 0; i < nlayers; ++i) {
      const bool even = ((last - i) & 1) == 0;  // gZ_i in P, X_i / dX_i in Q
      int& wg = even ? wa : wb;
      int& wx = even ? wb : wa;
      const int n8 = (layers[i].n + 7) / 8 * 8;
      wg = n8 > wg ? n8 : wg;
      wx = layers[i].k > wx ? layers[i].k : wx;
    }
    int64_t biggest = 0;
    for (int i = 0; i < nlayers; ++i) {
      const int64_t c = (int64_t)layers[i].n * layers[i].k + layers[i].n;
      biggest = c > biggest ? c : biggest;
    }
    d.sa = wa + 4;
    d.sb = wb + 4;
    int nsum = 0;
    for (int i = 0; i < nlayers; ++i) {
      d.db_off[i] = nsum;
      nsum += layers[i].n;
    }
    d.nsum = nsum;
    const int64_t tiles_floats = (int64_t)kWaves * 32 * (d.sa + d.sb);
    // the flush stages one layer's n*k + n sums over the tile region, which must hold it
    CTR_REQUIRE(biggest <= tiles_floats, CTR_ELIMIT);
    out->lds_bytes = sizeof(float) * (d.wfloats + tiles_floats + (int64_t)kWaves * nsum + kSlack);
  } else {
    for (int i = 0; i < nlayers; ++i) {
      int& win = (i & 1) ? wb : wa;   // layer i reads tile A (even i) / B (odd i) ...
      int& wout = (i & 1) ? wa : wb;  // ... and writes the other one
      win = layers[i].k > win ? layers[i].k : win;
      wout = layers[i].n > wout ? layers[i].n : wout;
    }
    d.sa = wa + 4;
    d.sb = wb + 4;
    out->lds_bytes = sizeof(float) * (d.wfloats + (int64_t)kWaves * 32 * (d.sa + d.sb) + kSlack);
  }
  out->slab = slab;
  CTR_REQUIRE(out->lds_bytes + sizeof(StackDesc) <= 160 * 1024, CTR_ELIMIT);
  return CTR_OK;
}

template <class S, bool BWD>
bool matches(const ctr_mlp_layer_t* layers, int nlayers, const StackDesc& d) {
  if (!S::kFixed || nlayers != S::kLayers) return false;
  using Y = Layout<S, BWD>;
  for (int i = 0; i < nlayers; ++i) {
    if (layers[i].n != S::N[i] || layers[i].k != S::K[i] || layers[i].act != S::ACT[i]) return false;
    if (d.l[i].w_off != Y::w_off(i) || d.l[i].b_off != Y::b_off(i) || d.l[i].acc_off != Y::acc_off(i)) return false;
    if (BWD && d.db_off[i] != Y::db_off(i)) return false;
  }
  return d.sa == Y::sa() && d.sb == Y::sb() && d.wfloats == Y::wfloats() && (!BWD || d.nsum == Y::nsum());
}

template <class K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return CTR_OK;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes) == hipSuccess
             ? CTR_OK
             : CTR_ELAUNCH;
}

}  // namespace

static int mlp_fwd_impl(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, int nlayers,
                        const ctr_mlp_head_t* head, void* stream) {
  CTR_REQUIRE(m >= 0, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  CTR_REQUIRE(x && ctr_aligned16(x) && ldx % 4 == 0, CTR_EALIGN);
  Built b;
  int rc = build(layers, nlayers, false, &b);
  if (rc != CTR_OK) return rc;
  CTR_REQUIRE(ldx >= b.d.l[0].k, CTR_EINVAL);
  const int64_t tiles = ctr_ceil_div(m, 32);
  int64_t grid = ctr_ceil_div(tiles, kWaves);
  if (grid > 256) grid = 256;  // persistent: one workgroup per CU, weights staged once
  HeadDesc hd{nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0};
  hipStream_t st = (hipStream_t)stream;
  if (head) {
    CTR_REQUIRE(head->w && head->c && head->out && head->p >= 0 && head->ldout >= 1, CTR_EINVAL);
    CTR_REQUIRE(head->act >= CTR_ACT_NONE && head->act <= CTR_ACT_SIGMOID, CTR_EINVAL);
    CTR_REQUIRE(head->p == 0 || (head->x && head->ldx >= head->p), CTR_EINVAL);
    CTR_REQUIRE(head->p <= 64 && head->p % 8 == 0 && head->p + layers[nlayers - 1].n <= kHeadMax, CTR_ELIMIT);
    CTR_REQUIRE(head->p == 0 || (ctr_aligned16(head->x) && head->ldx % 4 == 0), CTR_EALIGN);
    hd = HeadDesc{head->x, head->ldx, head->p, head->w, head->c, head->out, head->ldout, head->act};
    if (matches<NcfTowerShape, false>(layers, nlayers, b.d)) {
      // activations in matrix-core operand layout, sixteen samples per wave (mlp_mfma16.hip); CTR_MLP_16=0 keeps
      // the tile-walking kernels below
      static const bool m16 = [] { const char* e = getenv("CTR_MLP_16"); return !(e && e[0] == '0'); }();
      if (m16) {
        rc = ctr_ncf16_fwd(x, ldx, m, layers, head, st);
        if (rc != CTR_ELIMIT) return rc;
      }
      {  // two waves per SIMD (see mlp_fwd_direct_kernel)
        constexpr size_t bytes = DirectLayout<NcfTowerShape>::lds_bytes();
        static_assert(bytes + sizeof(StackDesc) + sizeof(float) * kHeadMax <= 160 * 1024, "eight strips + weights fit the CU");
        rc = allow_lds(mlp_fwd_direct_kernel<NcfTowerShape, true>, bytes);
        if (rc != CTR_OK) return rc;
        int64_t g8 = ctr_ceil_div(tiles, kWavesDirect);
        if (g8 > 256) g8 = 256;
        hipLaunchKernelGGL((mlp_fwd_direct_kernel<NcfTowerShape, true>), dim3((unsigned)g8), dim3(64 * kWavesDirect), bytes,
                           st, b.d, x, ldx, m, hd);
        return ctr_launch_status();
      }
    } else {
      rc = allow_lds(mlp_fwd_kernel<DynShape, true>, b.lds_bytes);
      if (rc != CTR_OK) return rc;
      hipLaunchKernelGGL((mlp_fwd_kernel<DynShape, true>), dim3((unsigned)grid), dim3(kThreads), b.lds_bytes, st, b.d, x,
                         ldx, m, hd);
    }
    return ctr_launch_status();
  }
  if (matches<NcfShape, false>(layers, nlayers, b.d)) {
    rc = allow_lds(mlp_fwd_kernel<NcfShape>, b.lds_bytes);
    if (rc != CTR_OK) return rc;
    hipLaunchKernelGGL(mlp_fwd_kernel<NcfShape>, dim3((unsigned)grid), dim3(kThreads), b.lds_bytes, st, b.d, x, ldx, m, hd);
  } else if (matches<NcfTowerShape, false>(layers, nlayers, b.d)) {
    rc = allow_lds(mlp_fwd_kernel<NcfTowerShape>, b.lds_bytes);
    if (rc != CTR_OK) return rc;
    hipLaunchKernelGGL(mlp_fwd_kernel<NcfTowerShape>, dim3((unsigned)grid), dim3(kThreads), b.lds_bytes, st, b.d, x, ldx, m,
                       hd);
  } else if (matches<DienAttShape, false>(layers, nlayers, b.d)) {
    rc = allow_lds(mlp_fwd_kernel<DienAttShape>, b.lds_bytes);
    if (rc != CTR_OK) return rc;
    if (2 * b.lds_bytes <= 160 * 1024) {  // two resident workgroups per CU: twice the waves to hide latency
      grid = ctr_ceil_div(tiles, kWaves);
      if (grid > 512) grid = 512;
    }
    hipLaunchKernelGGL(mlp_fwd_kernel<DienAttShape>, dim3((unsigned)grid), dim3(kThreads), b.lds_bytes, st, b.d, x, ldx, m,
                       hd);
  } else {
    rc = allow_lds(mlp_fwd_kernel<DynShape>, b.lds_bytes);
    if (rc != CTR_OK) return rc;
    hipLaunchKernelGGL(mlp_fwd_kernel<DynShape>, dim3((unsigned)grid), dim3(kThreads), b.lds_bytes, st, b.d, x, ldx, m, hd);
  }
  return ctr_launch_status();
}

extern "C" int ctr_mlp_fwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, int nlayers,
                           void* stream) {
  return mlp_fwd_impl(x, ldx, m, layers, nlayers, nullptr, stream);
}

extern "C" int ctr_mlp_head_fwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, int nlayers,
                                const ctr_mlp_head_t* head, void* stream) {
  CTR_REQUIRE(head != nullptr, CTR_EINVAL);
  return mlp_fwd_impl(x, ldx, m, layers, nlayers, head, stream);
}

// ctr_embed_fwd(fields -> out) followed by ctr_mlp_head_fwd(out[:, :k0], layers, head) in ONE launch, for the
// pattern the library has a kernel for (NeuralCF at BASELINE configs[1], mlp_mfma16.hip); CTR_ELIMIT otherwise:
// nothing was enqueued and the caller issues the two calls.
extern "C" int ctr_embed_mlp_head_fwd(const ctr_field_t* fields, int nfields, int64_t batch, float* out, int64_t ldo,
                                      int32_t* err_flag, int write_x, const ctr_mlp_layer_t* layers, int nlayers,
                                      const ctr_mlp_head_t* head, const ctr_head_fold_t* fold, void* stream) {
  CTR_REQUIRE(fields && nfields > 0 && nfields <= CTR_MAX_FIELDS && out && ldo > 0 && layers && nlayers > 0 && head,
              CTR_EINVAL);
  CTR_REQUIRE(batch >= 0, CTR_EINVAL);
  if (batch == 0) return CTR_OK;
  CTR_REQUIRE(head->w && head->c && head->out && head->ldout >= 1, CTR_EINVAL);
  CTR_REQUIRE(head->act >= CTR_ACT_NONE && head->act <= CTR_ACT_SIGMOID, CTR_EINVAL);
  static const bool m16 = [] { const char* e = getenv("CTR_MLP_16"); return !(e && e[0] == '0'); }();
  if (!m16 || nlayers != 4 || batch < 1024 || !ctr_aligned16(out) || ldo % 4 != 0) return CTR_ELIMIT;
  for (int l = 0; l < nlayers; ++l)
    if (!layers[l].w || !layers[l].y || !ctr_aligned16(layers[l].w)) return CTR_ELIMIT;
  return ctr_ncf16_gather_fwd(fields, nfields, batch, out, ldo, err_flag, write_x, layers, head, fold, (hipStream_t)stream);
}

extern "C" int ctr_mlp_bwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, int nlayers,
                           const float* gy, int64_t ldgy, float* gx, int64_t ldgx, float* workspace,
                           int64_t workspace_floats, void* stream) {
  CTR_REQUIRE(m >= 0, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  CTR_REQUIRE(x && gy && workspace, CTR_EINVAL);
  CTR_REQUIRE(ctr_aligned16(x) && ldx % 4 == 0, CTR_EALIGN);
  Built b;
  int rc = build(layers, nlayers, true, &b);
  if (rc != CTR_OK) return rc;
  CTR_REQUIRE(ldx >= b.d.l[0].k && ldgy >= b.d.l[nlayers - 1].n && (!gx || ldgx >= b.d.l[0].k), CTR_EINVAL);
  CTR_REQUIRE(b.d.ntiles <= 16, CTR_ELIMIT);  // dW accumulators must fit the register file
  const int maxt = b.d.ntiles <= 8 ? 8 : (b.d.ntiles <= 12 ? 12 : (b.d.ntiles <= 14 ? 14 : 16));
  const int64_t tiles = ctr_ceil_div(m, 32);
  int64_t grid = ctr_ceil_div(tiles, kWaves);
  if (grid > 256) grid = 256;
  CTR_REQUIRE(workspace_floats >= grid * b.slab, CTR_ELIMIT);
  hipStream_t st = (hipStream_t)stream;
#define CTR_LAUNCH_BWD(S, T)                                                                                       \
  do {                                                                                                             \
    rc = allow_lds(mlp_bwd_kernel<S, T>, b.lds_bytes);                                                             \
    if (rc != CTR_OK) return rc;                                                                                   \
    hipLaunchKernelGGL((mlp_bwd_kernel<S, T>), dim3((unsigned)grid), dim3(kThreads), b.lds_bytes, st, b.d, x, ldx, \
                       m, gy, ldgy, gx, ldgx, workspace, b.slab, HeadBwdDesc{});                                   \
  } while (0)
  if (matches<NcfShape, true>(layers, nlayers, b.d)) CTR_LAUNCH_BWD(NcfShape, 14);
  else if (matches<NcfTowerShape, true>(layers, nlayers, b.d)) CTR_LAUNCH_BWD(NcfTowerShape, 12);
  else if (matches<DienAttShape, true>(layers, nlayers, b.d)) {
    if (2 * b.lds_bytes <= 160 * 1024) {  // two resident workgroups per CU (the kernel is capped at 256 registers)
      grid = ctr_ceil_div(tiles, kWaves);
      if (grid > 512) grid = 512;
      CTR_REQUIRE(workspace_floats >= grid * b.slab, CTR_ELIMIT);
    }
    CTR_LAUNCH_BWD(DienAttShape, 6);
  }
  else if (maxt == 8) CTR_LAUNCH_BWD(DynShape, 8);
  else if (maxt == 12) CTR_LAUNCH_BWD(DynShape, 12);
  else if (maxt == 14) CTR_LAUNCH_BWD(DynShape, 14);
  else CTR_LAUNCH_BWD(DynShape, 16);
#undef CTR_LAUNCH_BWD
  rc = ctr_launch_status();
  if (rc != CTR_OK) return rc;
  CtrSegments segs;
  segs.n = 0;
  int64_t off = 0;
  for (int i = 0; i < nlayers; ++i) {
    const int64_t wn = (int64_t)layers[i].n * layers[i].k;
    segs.s[segs.n++] = CtrSegment{off, wn, layers[i].gw};
    segs.s[segs.n++] = CtrSegment{off + wn, layers[i].n, layers[i].gb};
    off += wn + layers[i].n;
  }
  return ctr_reduce_segments(workspace, (int)grid, b.slab, segs, st);
}

// fields != NULL: the stack's input is gathered from the fields' tables (ctr_embed_mlp_head_bwd); x then only stands for
// "the matrix the forward wrote the other columns into" and is not read
static int mlp_head_bwd_impl(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx, int64_t m,
                             const ctr_mlp_layer_t* layers, int nlayers, const ctr_mlp_head_grad_t* hg,
                             const ctr_head_fold_grad_t* fold, float* gx, int64_t ldgx, float* workspace,
                             int64_t workspace_floats, void* stream, float* zero_buf = nullptr, int64_t zero_floats = 0) {
  CTR_REQUIRE(m >= 0 && hg, CTR_EINVAL);
  // an empty batch still owes the caller the cleared gradient buffer (it passed zero_buf INSTEAD of filling it)
  if (m == 0) return zero_buf ? ctr_zero_fill(zero_buf, zero_floats, (hipStream_t)stream) : CTR_OK;
  CTR_REQUIRE(x && workspace, CTR_EINVAL);
  CTR_REQUIRE(ctr_aligned16(x) && ldx % 4 == 0, CTR_EALIGN);
  CTR_REQUIRE(hg->prob && hg->gprob && hg->x && hg->w && hg->gx && hg->gw && hg->gc, CTR_EINVAL);
  CTR_REQUIRE(hg->act >= CTR_ACT_NONE && hg->act <= CTR_ACT_SIGMOID && hg->ldprob >= 1 && hg->ldgprob >= 1, CTR_EINVAL);
  Built b;
  int rc = build(layers, nlayers, true, &b);
  if (rc != CTR_OK) return rc;
  CTR_REQUIRE(ldx >= b.d.l[0].k && (!gx || ldgx >= b.d.l[0].k), CTR_EINVAL);
  // only the pinned NeuralCF tower with a 64-column extra operand has this path
  if (!matches<NcfTowerShape, true>(layers, nlayers, b.d) || hg->p != kHeadBwdP) return CTR_ELIMIT;
  CTR_REQUIRE(hg->ldx >= kHeadBwdP && hg->ldgx >= kHeadBwdP, CTR_EINVAL);
  const ctr_mlp_layer_t& lastl = layers[nlayers - 1];
  CTR_REQUIRE(ctr_aligned16(hg->x) && hg->ldx % 4 == 0 && ctr_aligned16(hg->gx) && hg->ldgx % 4 == 0 &&
                  ctr_aligned16(lastl.y) && lastl.ldy % 4 == 0,
              CTR_EALIGN);
  const int64_t tiles = ctr_ceil_div(m, 32);
  int64_t grid = ctr_ceil_div(tiles, kWaves);
  if (grid > 256) grid = 256;
  const int64_t slab = b.slab + kHeadBwdSums;
  CTR_REQUIRE(workspace_floats >= grid * slab, CTR_ELIMIT);
  hipStream_t st = (hipStream_t)stream;
  // activations and gradients in matrix-core operand layout, sixteen samples per wave (mlp_mfma16.hip; same slab
  // layout); CTR_MLP_16=0 keeps the tile-walking kernels below
  static const bool m16 = [] { const char* e = getenv("CTR_MLP_16"); return !(e && e[0] == '0'); }();
  bool done16 = false;
  if (m16 && gx && slab == ctr_ncf16_slab_floats()) {
    int g16 = 0;
    rc = fields ? ctr_ncf16_gather_bwd(fields, nfields, m, layers, hg, gx, ldgx, workspace, workspace_floats, &g16, zero_buf,
                                       zero_floats, st)
                : ctr_ncf16_bwd(x, ldx, m, layers, hg, gx, ldgx, workspace, workspace_floats, &g16, st);
    if (rc == CTR_OK) {
      grid = g16;
      done16 = true;
    } else if (rc != CTR_ELIMIT) {
      return rc;
    }
  }
  if (fields && !done16) return CTR_ELIMIT;   // only the operand-layout kernel gathers
  const HeadBwdDesc hb{hg->gprob, hg->ldgprob, hg->prob, hg->ldprob, hg->x, hg->ldx, hg->w, hg->gx, hg->ldgx, hg->act};
  if (done16) {
    // launched above
  } else {
    if (m % 32 == 0) {  // every tile full (the BASELINE batch): the instantiation without row-range checks
      rc = allow_lds(mlp_bwd_kernel<NcfTowerShape, 12, true, true>, b.lds_bytes);
      if (rc != CTR_OK) return rc;
      hipLaunchKernelGGL((mlp_bwd_kernel<NcfTowerShape, 12, true, true>), dim3((unsigned)grid), dim3(kThreads),
                         b.lds_bytes, st, b.d, x, ldx, m, nullptr, 0, gx, ldgx, workspace, slab, hb);
    } else {
      rc = allow_lds(mlp_bwd_kernel<NcfTowerShape, 12, true>, b.lds_bytes);
      if (rc != CTR_OK) return rc;
      hipLaunchKernelGGL((mlp_bwd_kernel<NcfTowerShape, 12, true>), dim3((unsigned)grid), dim3(kThreads), b.lds_bytes,
                         st, b.d, x, ldx, m, nullptr, 0, gx, ldgx, workspace, slab, hb);
    }
  }
  rc = ctr_launch_status();
  if (rc != CTR_OK) return rc;
  CtrSegments segs;
  segs.n = 0;
  int64_t off = 0;
  for (int i = 0; i < nlayers; ++i) {
    const int64_t wn = (int64_t)layers[i].n * layers[i].k;
    segs.s[segs.n++] = CtrSegment{off, wn, layers[i].gw};
    segs.s[segs.n++] = CtrSegment{off + wn, layers[i].n, layers[i].gb};
    off += wn + layers[i].n;
  }
  if (fold) {
    if (fold->p != kHeadBwdP || fold->n != 64 || fold->k != kHeadBwdN || !fold->u_full || !fold->w || fold->ldw < fold->k ||
        (fold->gw && fold->ldgw < fold->k))
      return CTR_EINVAL;   // (unreachable from the C ABI: ctr_embed_mlp_head_bwd checks the same before anything is enqueued)
    const CtrHeadFoldGrad F{fold->u_full, fold->w, fold->ldw, fold->b, hg->gw, hg->gc, fold->gu_full, fold->gw, fold->ldgw,
                            fold->gb, fold->gb2};
    return ctr_reduce_segments_fold(workspace, (int)grid, slab, segs, off, F, st);
  }
  segs.s[segs.n++] = CtrSegment{off, kHeadBwdP + kHeadBwdN, hg->gw};
  segs.s[segs.n++] = CtrSegment{off + kHeadBwdP + kHeadBwdN, 1, hg->gc};
  return ctr_reduce_segments(workspace, (int)grid, slab, segs, st);
}

extern "C" int ctr_mlp_head_bwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, int nlayers,
                                const ctr_mlp_head_grad_t* hg, float* gx, int64_t ldgx, float* workspace,
                                int64_t workspace_floats, void* stream) {
  return mlp_head_bwd_impl(nullptr, 0, x, ldx, m, layers, nlayers, hg, nullptr, gx, ldgx, workspace, workspace_floats, stream);
}

// Backward of ctr_embed_mlp_head_fwd(..., write_x = 0, ...): as ctr_mlp_head_bwd, with the stack's input gathered again
// from the fields' tables by the samples' ids instead of read from memory.  CTR_ELIMIT: not the pattern.
extern "C" int ctr_embed_mlp_head_bwd(const ctr_field_t* fields, int nfields, int64_t batch, const ctr_mlp_layer_t* layers,
                                      int nlayers, const ctr_mlp_head_grad_t* hg, const ctr_head_fold_grad_t* fold,
                                      float* gx, int64_t ldgx, float* workspace, int64_t workspace_floats,
                                      float* zero_buf, int64_t zero_floats, void* stream) {
  CTR_REQUIRE(fields && nfields > 0 && nfields <= CTR_MAX_FIELDS && hg, CTR_EINVAL);
  CTR_REQUIRE(!zero_buf || zero_floats >= 0, CTR_EINVAL);
  if (fold)
    CTR_REQUIRE(fold->p == kHeadBwdP && fold->n == 64 && fold->k == kHeadBwdN && fold->u_full && fold->w &&
                    fold->ldw >= fold->k && (!fold->gw || fold->ldgw >= fold->k),
                CTR_EINVAL);
  return mlp_head_bwd_impl(fields, nfields, hg->x, hg->ldx, batch, layers, nlayers, hg, fold, gx, ldgx, workspace,
                           workspace_floats, stream, zero_buf, zero_floats);
}
