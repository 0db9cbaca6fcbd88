// Shared device/host helpers for libctrhip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ctrhip.h"

#define CTR_WAVE 64

#define CTR_REQUIRE(cond, code) \
  do {                          \
    if (!(cond)) return (code); \
  } while (0)

static inline int ctr_launch_status() { return hipGetLastError() == hipSuccess ? CTR_OK : CTR_ELAUNCH; }

static inline int64_t ctr_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

static inline bool ctr_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// memory-bound grids: enough blocks to fill 256 CUs x 8, grid-stride the rest (guide G11)
static inline int ctr_stream_grid(int64_t work_items, int per_block) {
  int64_t g = ctr_ceil_div(work_items, per_block);
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;
  return (int)g;
}

__device__ __forceinline__ float ctr_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over aligned groups of G lanes (G power of two <= 64)
template <int G>
__device__ __forceinline__ float ctr_group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float ctr_sigmoid(float z) { return 1.0f / (1.0f + expf(-z)); }

__device__ __forceinline__ float ctr_act(float z, int act) {
  if (act == CTR_ACT_RELU) return z > 0.0f ? z : 0.0f;
  if (act == CTR_ACT_SIGMOID) return ctr_sigmoid(z);
  return z;
}

// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float ctr_act_grad(float y, int act) {
  if (act == CTR_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
  if (act == CTR_ACT_SIGMOID) return y * (1.0f - y);
  return 1.0f;
}

// Pointers that reach a kernel through an LDS/kernarg descriptor are "generic" to
// the compiler and lower to flat_* instructions; these helpers assert the global
// address space so loads/stores/atomics become global_* (vmcnt only).
#define CTR_GLOBAL __attribute__((address_space(1)))
typedef float ctr_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float ctr_ldg(const float* p) { return *(const CTR_GLOBAL float*)p; }
__device__ __forceinline__ int64_t ctr_ldg(const int64_t* p) { return *(const CTR_GLOBAL int64_t*)p; }
__device__ __forceinline__ float4 ctr_ldg(const float4* p) {
  const ctr_f32x4 v = *(const CTR_GLOBAL ctr_f32x4*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void ctr_stg(float* p, float v) { *(CTR_GLOBAL float*)p = v; }
__device__ __forceinline__ void ctr_stg(float4* p, float4 v) {
  ctr_f32x4 w;
  w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
  *(CTR_GLOBAL ctr_f32x4*)p = w;
}
__device__ __forceinline__ void ctr_atomic_add_global(float* p, float v) {
  (void)__builtin_amdgcn_global_atomic_fadd_f32((CTR_GLOBAL float*)p, v);
}

// Unsigned division by a run-time constant without the ~40-instruction u64/u32
// software divide (Hacker's Delight 10-9 / libdivide "branchfree" form), exact for
// every 32-bit numerator:  q = (t + ((n - t) >> s1)) >> s2,  t = mulhi(m, n).
struct CtrFastDiv {
  uint32_t d, m, s1, s2;
};
static inline CtrFastDiv ctr_fastdiv(uint32_t d) {
  CtrFastDiv f;
  f.d = d;
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  f.m = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  f.s1 = s < 1 ? s : 1;
  f.s2 = s > 0 ? s - 1 : 0;
  return f;
}
__device__ __forceinline__ uint32_t ctr_div(uint32_t n, const CtrFastDiv& f) {
  const uint32_t t = __umulhi(f.m, n);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

// internal: second pass over per-workgroup partials (reduce.hip)
#define CTR_MAX_SEGMENTS 40
struct CtrSegment {
  int64_t off;    // first float of the segment inside one partial
  int64_t count;  // floats
  float* dst;     // accumulated into (+=)
};
struct CtrSegments {
  int n;
  CtrSegment s[CTR_MAX_SEGMENTS];
};
int ctr_reduce_segments(const float* ws, int parts, int64_t stride, const CtrSegments& segs, hipStream_t st);
// the same plus NeuralCF's head sums at float `head_off` of a partial (72 weights + 1 bias) and ctr_fold_head_bwd's chain
// rule from their totals (p = 64, n = 64, k = 8), in one launch
struct CtrHeadFoldGrad {
  const float* u_full; const float* w; int64_t ldw; const float* b;
  float* gwfold; float* gcfold;          // receive the sums (+=)
  float* gu_full; float* gw; int64_t ldgw; float* gb; float* gb2;   // each nullable, accumulated (+=)
};
int ctr_reduce_segments_fold(const float* ws, int parts, int64_t stride, const CtrSegments& segs, int64_t head_off,
                             const CtrHeadFoldGrad& fold, hipStream_t st);

int ctr_zero_fill(float* p, int64_t n, hipStream_t st);   // reduce.hip: p 16-byte aligned, n % 4 == 0

// mlp_mfma16.hip: the pinned NeuralCF tower + 64-column head with activations in matrix-core operand layout
// (CTR_ELIMIT: shape / alignment not taken, nothing enqueued)
int ctr_ncf16_fwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, const ctr_mlp_head_t* head,
                  hipStream_t st);
int ctr_ncf16_gather_fwd(const ctr_field_t* fields, int nfields, int64_t m, float* out, int64_t ldo, int32_t* err_flag,
                         int write_x, const ctr_mlp_layer_t* layers, const ctr_mlp_head_t* head, const ctr_head_fold_t* fold,
                         hipStream_t st);
int ctr_ncf16_gather_bwd(const ctr_field_t* fields, int nfields, int64_t m, const ctr_mlp_layer_t* layers,
                         const ctr_mlp_head_grad_t* hg, float* gx, int64_t ldgx, float* workspace,
                         int64_t workspace_floats, int* grid_out, float* zero_buf, int64_t zero_floats, hipStream_t st);
int ctr_ncf16_bwd(const float* x, int64_t ldx, int64_t m, const ctr_mlp_layer_t* layers, const ctr_mlp_head_grad_t* hg,
                  float* gx, int64_t ldgx, float* workspace, int64_t workspace_floats, int* grid_out, hipStream_t st);
int ctr_ncf16_slab_floats();
// internal (not part of the C ABI): single-output-unit linear layer, linear_n1.hip
bool ctr_n1_supported(int k);
// embed_sorted.hip: sorted segmented-reduce backward for small tables (see there)
int ctr_embed_bwd_sorted(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx, int64_t batch,
                         const float* gout, int64_t ldo, float* workspace, int64_t workspace_floats,
                         int64_t* used_floats, unsigned char* handled, hipStream_t st);
// embed_bag.hip: bag-table gradients by register accumulation per output column (see there)
int ctr_embed_bwd_bags(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx, int64_t batch,
                       const float* gout, int64_t ldo, float* workspace, int64_t workspace_floats,
                       int64_t* used_floats, unsigned char* handled, hipStream_t st);
// embed_bag.hip: forward of the wide bag fields in a kernel of their own (see there)
int ctr_embed_fwd_bags(const ctr_field_t* fields, int nfields, const float* x, int64_t ldx, int64_t batch, float* out,
                       int64_t ldo, unsigned char* handled, hipStream_t st);
// linear_skinny.hip: weight gradient of a layer with few units on both sides over a very long batch
bool ctr_skinny_dw_ok(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy,
                      const float* gw, int64_t ldgw, int64_t m, int n, int k, int act);
int ctr_skinny_dw(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy, float* gw,
                  float* gb, int64_t m, int n, int k, int act, float* workspace, int64_t workspace_floats,
                  hipStream_t st);
// gemm_dlds.hip: forward GEMM with direct global->LDS operand loads
// wide layers on a long batch: 256 x 256 macro tile, one workgroup per CU (gemm_wide.hip)
bool ctr_gemm_wide_ok(const float* x, int64_t ldx, const float* w, int64_t ldw, int64_t m, int n, int k);
int ctr_gemm_wide_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy,
                      int64_t m, int n, int k, int act, hipStream_t st);
bool ctr_gemm_dlds_ok(const float* x, int64_t ldx, const float* w, int64_t ldw, int64_t m, int n, int k);
int ctr_gemm_dlds_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* res,
                      int64_t ldr, float* y, int64_t ldy, int64_t m, int n, int k, int act, hipStream_t st);
int ctr_gemm_dlds_fwd_group(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* res,
                            int64_t ldr, int group, float* y, int64_t ldy, int64_t m, int n, int k, int act,
                            hipStream_t st, uint32_t* mask = nullptr, int64_t ldmask = 0);
// gemm_dlds_dw.hip: weight gradient with direct global->LDS operand loads
bool ctr_gemm_dlds_dw_ok(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy,
                         const float* gw, int64_t ldgw, int64_t m, int n, int k, int act);
int ctr_gemm_dlds_dw(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gy, int64_t ldgy, float* gw,
                     float* gb, int64_t m, int n, int k, int act, float* workspace, int64_t workspace_floats,
                     hipStream_t st);
// gemm_dlds_dx.hip: input gradient with direct global->LDS operand loads
bool ctr_gemm_dlds_dx_ok(const float* w, int64_t ldw, const float* y, int64_t ldy, const float* gy, int64_t ldgy,
                         int64_t m, int n, int k, int act);
int ctr_gemm_dlds_dx(const float* w, int64_t ldw, const float* y, int64_t ldy, const float* gy, int64_t ldgy, float* gx,
                     int64_t ldgx, int accumulate, int64_t m, int n, int k, int act, hipStream_t st);
int ctr_n1_fwd(const float* x, int64_t ldx, const float* w, const float* bias, const float* res, int64_t ldr, float* y,
               int64_t ldy, int64_t m, int k, int act, hipStream_t st);
int ctr_n1_bwd(const float* x, int64_t ldx, const float* w, const float* y, int64_t ldy, const float* gy, int64_t ldgy,
               float* gx, int64_t ldgx, int accumulate_gx, float* gw, float* gb, int64_t m, int k, int act,
               float* ws, int64_t ws_floats, hipStream_t st, int act_in = CTR_ACT_NONE);

