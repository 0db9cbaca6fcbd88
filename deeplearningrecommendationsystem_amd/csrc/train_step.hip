// The two non-model pieces of a Trainer.train_loop body (trainer/trainer.py:37-39):
//   * torch.nn.BCELoss() (every script, e.g. scripts/pnn.py:54): mean over the batch of
//     -[y*max(log p,-100) + (1-y)*max(log(1-p),-100)], and its gradient
//     (p - y) / max(p(1-p), 1e-12) * gloss / n          -- ATen's formulas, one pass each;
//   * torch.optim.Adam(lr, betas, eps, weight_decay) (e.g. scripts/pnn.py:55): dense Adam
//     with L2 decay over every parameter, all tensors in ONE launch, 4 reads + 3 writes
//     per element (g, p, m, v -> p, m, v): HBM-bound, 28 B per parameter.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;

// One launch: every workgroup stores its partial, takes a ticket, and the workgroup that draws the last
// ticket sums the partials in index order (bitwise reproducible) and re-arms the ticket.  (Three
// launches before -- memset, partials, reduction -- at ~4.5 us each on a 220 us step.)
__global__ void __launch_bounds__(kBlock)
bce_fwd_kernel(const float* __restrict__ p, int64_t ldp, const float* __restrict__ y, int64_t ldy, int64_t n,
               float inv_n, float* __restrict__ partial, unsigned int* __restrict__ ticket, float* __restrict__ loss,
               float* __restrict__ gp1 /* nullable: d loss / d p for an upstream gradient of exactly 1 */) {
  __shared__ float s_red[kBlock / 64];
  __shared__ bool s_last;
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float pv = p[i * ldp], yv = y[i * ldy];
    const float lp = fmaxf(logf(pv), -100.0f), l1p = fmaxf(logf(1.0f - pv), -100.0f);
    acc -= yv * lp + (1.0f - yv) * l1p;
    if (gp1) gp1[i] = (pv - yv) / fmaxf((1.0f - pv) * pv, 1e-12f) * inv_n;  // bce_bwd_kernel with gloss = 1
  }
  acc = ctr_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int w = 0; w < kBlock / 64; ++w) t += s_red[w];
    __hip_atomic_store(partial + blockIdx.x, t * inv_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // release my partial, acquire everybody else's if I am last
    const unsigned int drawn = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = drawn == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  // gridDim.x <= 256 = kBlock partials: one per thread, summed in a fixed tree
  float v = threadIdx.x < gridDim.x
                ? __hip_atomic_load(partial + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                : 0.0f;
  v = ctr_wave_sum(v);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int w = 0; w < kBlock / 64; ++w) t += s_red[w];
    loss[0] = t;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void __launch_bounds__(kBlock)
bce_bwd_kernel(const float* __restrict__ p, int64_t ldp, const float* __restrict__ y, int64_t ldy, int64_t n,
               const float* __restrict__ gloss, float inv_n, float* __restrict__ gp, int64_t ldg) {
  const float scale = gloss[0] * inv_n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float pv = p[i * ldp], yv = y[i * ldy];
    gp[i * ldg] = (pv - yv) / fmaxf((1.0f - pv) * pv, 1e-12f) * scale;
  }
}

struct AdamPack {
  ctr_adam_tensor_t t[CTR_ADAM_MAX_TENSORS];
  int n;
};

__global__ void __launch_bounds__(kBlock)
adam_kernel(const AdamPack P, float lr, float beta2, float omb1, float omb2, float eps, float weight_decay, float bc1,
            float bc2_sqrt) {  // omb = 1 - beta, formed in double on the host as torch does
  const ctr_adam_tensor_t t = P.t[blockIdx.y];
  const float step_size = lr / bc1;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < t.numel; i += (int64_t)gridDim.x * blockDim.x * 4) {
    if (i + 3 < t.numel) {
      float4 g = *reinterpret_cast<const float4*>(t.grad + i);
      float4 p = *reinterpret_cast<const float4*>(t.param + i);
      float4 m = *reinterpret_cast<const float4*>(t.exp_avg + i);
      float4 v = *reinterpret_cast<const float4*>(t.exp_avg_sq + i);
      float* gg = &g.x; float* pp = &p.x; float* mm = &m.x; float* vv = &v.x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gr = fmaf(weight_decay, pp[e], gg[e]);     // grad.add(param, alpha=wd)
        mm[e] = fmaf(omb1, gr - mm[e], mm[e]);           // exp_avg.lerp_(grad, 1 - beta1)
        vv[e] = vv[e] * beta2 + (omb2 * gr) * gr;      // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
        pp[e] -= step_size * (mm[e] / denom);
      }
      *reinterpret_cast<float4*>(t.param + i) = p;
      *reinterpret_cast<float4*>(t.exp_avg + i) = m;
      *reinterpret_cast<float4*>(t.exp_avg_sq + i) = v;
    } else {
      for (int64_t j = i; j < t.numel; ++j) {
        const float gr = fmaf(weight_decay, t.param[j], t.grad[j]);
        const float m = fmaf(omb1, gr - t.exp_avg[j], t.exp_avg[j]);
        const float v = t.exp_avg_sq[j] * beta2 + (omb2 * gr) * gr;
        t.exp_avg[j] = m;
        t.exp_avg_sq[j] = v;
        t.param[j] -= step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
      }
    }
  }
}

}  // namespace

extern "C" int ctr_bce_fwd(const float* prob, int64_t ldp, const float* target, int64_t ldt, int64_t n, float* loss,
                           float* workspace, int64_t workspace_floats, unsigned int* ticket, float* gprob_unit,
                           void* stream) {
  CTR_REQUIRE(n > 0 && prob && target && loss && workspace && ticket && ldp >= 1 && ldt >= 1, CTR_EINVAL);
  int64_t grid = ctr_ceil_div(n, kBlock * 4);
  if (grid > 256) grid = 256;
  CTR_REQUIRE(workspace_floats >= grid, CTR_ELIMIT);
  hipLaunchKernelGGL(bce_fwd_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, prob, ldp, target, ldt,
                     n, 1.0f / (float)n, workspace, ticket, loss, gprob_unit);
  return ctr_launch_status();
}

extern "C" int ctr_bce_bwd(const float* prob, int64_t ldp, const float* target, int64_t ldt, int64_t n,
                           const float* gloss, float* gprob, int64_t ldg, void* stream) {
  CTR_REQUIRE(n > 0 && prob && target && gloss && gprob && ldp >= 1 && ldt >= 1 && ldg >= 1, CTR_EINVAL);
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(ctr_stream_grid(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, prob, ldp,
                     target, ldt, n, gloss, 1.0f / (float)n, gprob, ldg);
  return ctr_launch_status();
}

extern "C" int ctr_adam_step(const ctr_adam_tensor_t* tensors, int ntensors, double lr, double beta1, double beta2,
                             double eps, double weight_decay, int64_t step, void* stream) {
  CTR_REQUIRE(tensors && ntensors >= 0 && step >= 1, CTR_EINVAL);
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < ntensors; base += CTR_ADAM_MAX_TENSORS) {
    AdamPack P;
    P.n = ntensors - base < CTR_ADAM_MAX_TENSORS ? ntensors - base : CTR_ADAM_MAX_TENSORS;
    int64_t longest = 0;
    for (int i = 0; i < P.n; ++i) {
      const ctr_adam_tensor_t& t = tensors[base + i];
      CTR_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq && t.numel >= 0, CTR_EINVAL);
      CTR_REQUIRE(ctr_aligned16(t.param) && ctr_aligned16(t.grad) && ctr_aligned16(t.exp_avg) &&
                      ctr_aligned16(t.exp_avg_sq),
                  CTR_EALIGN);
      P.t[i] = t;
      longest = t.numel > longest ? t.numel : longest;
    }
    if (longest == 0) continue;
    int64_t gx = ctr_ceil_div(longest, kBlock * 4);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)gx, (unsigned)P.n), dim3(kBlock), 0, st, P, (float)lr,
                       (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay,
                       (float)bc1, (float)sqrt(bc2));
    int rc = ctr_launch_status();
    if (rc != CTR_OK) return rc;
  }
  return CTR_OK;
}
