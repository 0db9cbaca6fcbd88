// nn.Linear with ONE output unit (the sigmoid heads, DIN's attention score layer,
// DeepFM's last deep layer): y[m] = act(x[m,:] . w + b (+ r[m])).  A 32-wide MFMA
// tile would idle 31/32 of the matrix core and, worse, re-read nothing: this is a
// streaming dot product, HBM-bound on x.  LPR lanes share a row (dwordx4 each when
// the rows are 16-B aligned), a workgroup walks rows with a grid stride.
// Backward in one pass over x: gx[m,:] = gz[m] w,  gw += sum_m gz[m] x[m,:],
// gb += sum_m gz[m]; per-lane register partials, one LDS reduction per workgroup,
// then k+1 global atomics per workgroup.
#include "ctr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxChunks = 8;  // k-chunks a lane may own: k <= 64 * 4 * 8

struct N1Args {
  const float* x; int64_t ldx;
  const float* w;
  const float* bias;
  const float* res; int64_t ldr;
  float* y; int64_t ldy;
  int64_t m; int k; int act; int lpr;
};

// U > 1: the row fits a single chunk per lane (k <= lpr * VEC) and U rows are in flight per lane group --
// one row at a time is a load-use loop that leaves the kernel latency-bound on long batches.
template <int VEC, int U>
__global__ void __launch_bounds__(kBlock) n1_fwd_kernel(const N1Args a) {
  const int sub = threadIdx.x % a.lpr;
  const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / a.lpr;
  const int64_t groups = ((int64_t)gridDim.x * blockDim.x) / a.lpr;
  if constexpr (U > 1) {
    const int c = sub * VEC;
    const bool live = c < a.k;
    float wv[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) wv[v] = live ? a.w[c + v] : 0.0f;
    const float bias = a.bias ? a.bias[0] : 0.0f;
    for (int64_t r0 = gid; r0 < a.m; r0 += groups * U) {
      float xv[U][VEC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * groups;
        const int64_t rc = r < a.m ? r : a.m - 1;
#pragma unroll
        for (int v = 0; v < VEC; ++v) xv[u][v] = 0.0f;
        if (live) {
          if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(a.x + rc * a.ldx + c);
            xv[u][0] = t.x; xv[u][1] = t.y; xv[u][2] = t.z; xv[u][VEC - 1] = t.w;
          } else {
            xv[u][0] = a.x[rc * a.ldx + c];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * groups;
        float acc = 0.0f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc = fmaf(xv[u][v], wv[v], acc);
        for (int o = a.lpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (sub == 0 && r < a.m) {
          float z = acc + bias;
          if (a.res) z += a.res[r * a.ldr];
          a.y[r * a.ldy] = ctr_act(z, a.act);
        }
      }
    }
    return;
  }
  for (int64_t r = gid; r < a.m; r += groups) {
    const float* xr = a.x + r * a.ldx;
    float acc = 0.0f;
    for (int c = sub * VEC; c < a.k; c += a.lpr * VEC) {
      if (VEC == 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xr + c);
        const float4 wv = *reinterpret_cast<const float4*>(a.w + c);
        acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
      } else {
        acc = fmaf(xr[c], a.w[c], acc);
      }
    }
    for (int o = a.lpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (sub == 0) {
      float z = acc;
      if (a.bias) z += a.bias[0];
      if (a.res) z += a.res[r * a.ldr];
      a.y[r * a.ldy] = ctr_act(z, a.act);
    }
  }
}

struct N1BwdArgs {
  const float* x; int64_t ldx;
  const float* w;
  const float* y; int64_t ldy;
  const float* gy; int64_t ldgy;
  float* gx; int64_t ldgx; int accumulate_gx;
  float* gw; float* gb;
  int64_t m; int k; int act; int lpr;
  float* ws;  // per-workgroup partials [gridDim.x][k+1] (weights then bias), or NULL -> atomics
  int act_in;  // != NONE: gx *= act_in'(x) -- x is the previous layer's activation output (gx may be x itself)
};

// U > 1: the row needs a single chunk per lane (k <= lpr * VEC) and U rows are in flight per
// lane -- a load-use loop over one row at a time leaves this kernel latency-bound at ~40 % of
// the HBM rate.
template <int VEC, int U>
__global__ void __launch_bounds__(kBlock) n1_bwd_kernel(const N1BwdArgs a) {
  __shared__ float s_red[kBlock * VEC];
  const int sub = threadIdx.x % a.lpr;
  const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / a.lpr;
  const int64_t groups = ((int64_t)gridDim.x * blockDim.x) / a.lpr;
  constexpr int kChunks = U > 1 ? 1 : kMaxChunks;
  float wacc[kChunks][VEC];
#pragma unroll
  for (int q = 0; q < kChunks; ++q)
#pragma unroll
    for (int v = 0; v < VEC; ++v) wacc[q][v] = 0.0f;
  float bacc = 0.0f;
  const bool need_w = a.gw != nullptr;
  if constexpr (U > 1) {
    const int c = sub * VEC;
    const bool live = c < a.k;
    float wv[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) wv[v] = (live && a.gx) ? a.w[c + v] : 0.0f;
    for (int64_t r0 = gid; r0 < a.m; r0 += groups * U) {
      float gz[U], xv[U][VEC], pv[U][VEC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * groups;
        const bool ok = r < a.m;
        const int64_t rc = ok ? r : a.m - 1;
        gz[u] = a.gy[rc * a.ldgy];
        if (a.act != CTR_ACT_NONE) gz[u] *= ctr_act_grad(a.y[rc * a.ldy], a.act);
        if (!ok) gz[u] = 0.0f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) xv[u][v] = pv[u][v] = 0.0f;
        if (live && (need_w || a.act_in != CTR_ACT_NONE)) {
          if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(a.x + rc * a.ldx + c);
            xv[u][0] = t.x; xv[u][1] = t.y; xv[u][2] = t.z; xv[u][VEC - 1] = t.w;
          } else {
            xv[u][0] = a.x[rc * a.ldx + c];
          }
        }
        if (live && a.gx && a.accumulate_gx) {
          if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(a.gx + rc * a.ldgx + c);
            pv[u][0] = t.x; pv[u][1] = t.y; pv[u][2] = t.z; pv[u][VEC - 1] = t.w;
          } else {
            pv[u][0] = a.gx[rc * a.ldgx + c];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * groups;
        if (sub == 0) bacc += gz[u];
#pragma unroll
        for (int v = 0; v < VEC; ++v) wacc[0][v] = fmaf(gz[u], xv[u][v], wacc[0][v]);
        if (live && a.gx && r < a.m) {
          float o[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            float t = gz[u] * wv[v];
            if (a.act_in != CTR_ACT_NONE) t *= ctr_act_grad(xv[u][v], a.act_in);
            o[v] = t + pv[u][v];
          }
          if (VEC == 4)
            *reinterpret_cast<float4*>(a.gx + r * a.ldgx + c) = make_float4(o[0], o[1], o[2], o[VEC - 1]);
          else
            a.gx[r * a.ldgx + c] = o[0];
        }
      }
    }
  } else {
  for (int64_t r = gid; r < a.m; r += groups) {
    float gz = a.gy[r * a.ldgy];
    if (a.act != CTR_ACT_NONE) gz *= ctr_act_grad(a.y[r * a.ldy], a.act);
    if (sub == 0) bacc += gz;
    const float* xr = a.x ? a.x + r * a.ldx : nullptr;
    float* gxr = a.gx ? a.gx + r * a.ldgx : nullptr;
#pragma unroll
    for (int q = 0; q < kChunks; ++q) {
      const int c = (sub + q * a.lpr) * VEC;
      if (c < a.k) {
        if (VEC == 4) {
          if (need_w) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + c);
            wacc[q][0] = fmaf(gz, xv.x, wacc[q][0]); wacc[q][1] = fmaf(gz, xv.y, wacc[q][1]);
            wacc[q][2] = fmaf(gz, xv.z, wacc[q][2]); wacc[q][VEC - 1] = fmaf(gz, xv.w, wacc[q][VEC - 1]);
          }
          if (gxr) {
            const float4 wv = *reinterpret_cast<const float4*>(a.w + c);
            float4 o = make_float4(gz * wv.x, gz * wv.y, gz * wv.z, gz * wv.w);
            if (a.accumulate_gx) {
              const float4 p = *reinterpret_cast<const float4*>(gxr + c);
              o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
            }
            *reinterpret_cast<float4*>(gxr + c) = o;
          }
        } else {
          if (need_w) wacc[q][0] = fmaf(gz, xr[c], wacc[q][0]);
          if (gxr) gxr[c] = a.accumulate_gx ? gxr[c] + gz * a.w[c] : gz * a.w[c];
        }
      }
    }
  }
  }
  // reduce the per-group partials of the workgroup, chunk by chunk
  const int ngroups = kBlock / a.lpr;
  if (need_w) {
#pragma unroll
    for (int q = 0; q < kChunks; ++q) {
      if (q * a.lpr * VEC >= a.k) break;
      __syncthreads();
#pragma unroll
      for (int v = 0; v < VEC; ++v) s_red[threadIdx.x * VEC + v] = wacc[q][v];
      __syncthreads();
      if (threadIdx.x < a.lpr) {
        const int c = (threadIdx.x + q * a.lpr) * VEC;
        if (c < a.k) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            float t = 0.0f;
            for (int g = 0; g < ngroups; ++g) t += s_red[(g * a.lpr + threadIdx.x) * VEC + v];
            if (a.ws)
              a.ws[(int64_t)blockIdx.x * (a.k + 1) + c + v] = t;
            else if (t != 0.0f)
              unsafeAtomicAdd(a.gw + c + v, t);
          }
        }
      }
    }
  }
  if (a.gb || a.ws) {
    __syncthreads();
    s_red[threadIdx.x] = bacc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.0f;
      for (int g = 0; g < kBlock; ++g) t += s_red[g];
      if (a.ws)
        a.ws[(int64_t)blockIdx.x * (a.k + 1) + a.k] = t;
      else if (t != 0.0f)
        unsafeAtomicAdd(a.gb, t);
    }
  }
}

inline int pow2_ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

bool ctr_n1_supported(int k) { return k <= 64 * kMaxChunks; }

int ctr_n1_fwd(const float* x, int64_t ldx, const float* w, const float* bias, const float* res, int64_t ldr, float* y,
               int64_t ldy, int64_t m, int k, int act, hipStream_t st) {
  const bool vec = k % 4 == 0 && ldx % 4 == 0 && ctr_aligned16(x) && ctr_aligned16(w);
  const int units = vec ? k / 4 : k;
  int lpr = pow2_ceil(units);
  if (lpr > 64) lpr = 64;
  N1Args a{x, ldx, w, bias, res, ldr, y, ldy, m, k, act, lpr};
  // one chunk per lane and a long batch: four rows in flight per lane group
  const bool single = (vec ? 4 : 1) * lpr >= k && m >= 262144;
  const int grid = ctr_stream_grid(single ? ctr_ceil_div(m * lpr, 4) : m * lpr, kBlock);
  if (vec && single) hipLaunchKernelGGL((n1_fwd_kernel<4, 4>), dim3(grid), dim3(kBlock), 0, st, a);
  else if (vec) hipLaunchKernelGGL((n1_fwd_kernel<4, 1>), dim3(grid), dim3(kBlock), 0, st, a);
  else if (single) hipLaunchKernelGGL((n1_fwd_kernel<1, 4>), dim3(grid), dim3(kBlock), 0, st, a);
  else hipLaunchKernelGGL((n1_fwd_kernel<1, 1>), dim3(grid), dim3(kBlock), 0, st, a);
  return ctr_launch_status();
}

int ctr_n1_bwd(const float* x, int64_t ldx, const float* w, const float* y, int64_t ldy, const float* gy, int64_t ldgy,
               float* gx, int64_t ldgx, int accumulate_gx, float* gw, float* gb, int64_t m, int k, int act,
               float* ws, int64_t ws_floats, hipStream_t st, int act_in) {
  const bool vec = k % 4 == 0 && (!x || (ldx % 4 == 0 && ctr_aligned16(x))) && (!w || ctr_aligned16(w)) &&
                   (!gx || (ldgx % 4 == 0 && ctr_aligned16(gx)));
  const int units = vec ? k / 4 : k;
  int lpr = pow2_ceil(units);
  if (lpr > 64) lpr = 64;
  if ((vec ? 4 : 1) * lpr * kMaxChunks < k) return CTR_ELIMIT;
  int grid = ctr_stream_grid(m * lpr, kBlock);
  if (grid > 1024) grid = 1024;
  const bool slabs = gw && ws && ws_floats >= (int64_t)grid * (k + 1) && grid > 8;
  if ((gw || gb) && !slabs && grid > 128) grid = 128;  // same-address atomics serialise: keep the chains short
  N1BwdArgs a{x, ldx, w, y, ldy, gy, ldgy, gx, ldgx, accumulate_gx, gw, gb, m, k, act, lpr, slabs ? ws : nullptr,
              act_in};
  const bool single = (vec ? 4 : 1) * lpr >= k && x && w;  // one chunk per lane: 4 rows in flight
  if (act_in != CTR_ACT_NONE && !single) return CTR_ELIMIT;  // the masked form exists for one chunk per lane only
  if (vec && single)
    hipLaunchKernelGGL((n1_bwd_kernel<4, 4>), dim3(grid), dim3(kBlock), 0, st, a);
  else if (vec)
    hipLaunchKernelGGL((n1_bwd_kernel<4, 1>), dim3(grid), dim3(kBlock), 0, st, a);
  else if (single)
    hipLaunchKernelGGL((n1_bwd_kernel<1, 4>), dim3(grid), dim3(kBlock), 0, st, a);
  else
    hipLaunchKernelGGL((n1_bwd_kernel<1, 1>), dim3(grid), dim3(kBlock), 0, st, a);
  int rc = ctr_launch_status();
  if (rc != CTR_OK || !slabs) return rc;
  CtrSegments segs;
  segs.n = gb ? 2 : 1;
  segs.s[0] = CtrSegment{0, k, gw};
  segs.s[1] = CtrSegment{k, 1, gb};
  return ctr_reduce_segments(ws, grid, k + 1, segs, st);
}

// C ABI (include/ctrhip.h): backward of a single-unit layer y = x w^T + b whose input x is the previous layer's
// activation output, with that activation's derivative folded in:
//   gx = (gy w) * act_in'(x)   (gx may be x itself: every element is read, then written, by the same lane)
//   gw += gy^T x,  gb += sum gy
extern "C" int ctr_linear_n1_bwd_masked(const float* x, int64_t ldx, const float* w, const float* gy, int64_t ldgy,
                                        int act_in, float* gx, int64_t ldgx, float* gw, float* gb, int64_t m, int k,
                                        float* workspace, int64_t workspace_floats, void* stream) {
  CTR_REQUIRE(m >= 0 && k >= 1, CTR_EINVAL);
  if (m == 0) return CTR_OK;
  CTR_REQUIRE(x && w && gy && gx && ldx >= k && ldgx >= k && ldgy >= 1, CTR_EINVAL);
  CTR_REQUIRE(act_in >= CTR_ACT_NONE && act_in <= CTR_ACT_SIGMOID, CTR_EINVAL);
  CTR_REQUIRE(ctr_n1_supported(k), CTR_ELIMIT);
  return ctr_n1_bwd(x, ldx, w, nullptr, 0, gy, ldgy, gx, ldgx, 0, gw, gb, m, k, CTR_ACT_NONE, workspace,
                    workspace_floats, (hipStream_t)stream, act_in);
}
