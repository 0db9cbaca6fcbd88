// dev tool (not product): what random 64..256-byte row gathers / scatters can reach on MI355X, as a
// function of where the rows live (L2 / Infinity Cache / HBM), of the id order (sample order vs
// sorted per table) and of the id distribution (uniform / Zipf).  One line per variant.
//   hipcc -O3 --offload-arch=gfx950 dev/gather_ceiling.hip -o dev/build/gather_ceiling
//   dev/build/gather_ceiling [quick]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <random>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL __attribute__((address_space(1)))

constexpr int kBlock = 256;

__global__ void fill_kernel(float* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed;
    h ^= h >> 15;
    p[i] = (float)(h & 0xffff) * (1.0f / 65536.0f);
  }
}

// item i = (sample b, field f), sample-major.  mode: 0 copy, 1 read only, 2 write only
// src row = tab + f*V*E + idx[i]*E ; dst = out + dst_item*E  (dst_item = i, or perm[i] for sorted lists)
template <int UNROLL, int MODE, bool PERM, typename IdxT>
__global__ void __launch_bounds__(kBlock)
rows_kernel(const float* __restrict__ tab, int64_t V, int E, int F, const IdxT* __restrict__ idx,
            const uint32_t* __restrict__ perm, const uint32_t* __restrict__ field_of, uint32_t items,
            float* __restrict__ out, float* __restrict__ sink) {
  const int lpr = E / 4;
  const int sub = threadIdx.x % lpr;
  const uint32_t rpb = kBlock / lpr;
  const uint32_t per_block = rpb * UNROLL;
  const uint32_t slot = threadIdx.x / lpr;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (uint32_t base = blockIdx.x * per_block; base < items; base += gridDim.x * per_block) {
    const float* src[UNROLL];
    float* dst[UNROLL];
    bool live[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      const uint32_t i = base + slot + k * rpb;
      live[k] = i < items;
      const uint32_t ii = live[k] ? i : 0;
      uint32_t f, d;
      if (PERM) {
        f = field_of[ii];
        d = perm[ii];
      } else {
        f = ii % (uint32_t)F;
        d = ii;
      }
      const int64_t r = (MODE == 2) ? 0 : (int64_t)idx[ii];
      src[k] = tab + ((int64_t)f * V + r) * E + sub * 4;
      dst[k] = out + (int64_t)d * E + sub * 4;
    }
    f32x4 v[UNROLL];
    if (MODE != 2) {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) v[k] = *(const GLOBAL f32x4*)(src[k]);
    } else {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) v[k] = acc + (float)k;
    }
    if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) acc += v[k];
    } else {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k)
        if (live[k]) *(GLOBAL f32x4*)(dst[k]) = v[k];
    }
  }
  if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == -12345.678f) sink[threadIdx.x] = acc.x;
}

// the same copy with the row loads issued from inline asm so that the cache-policy bits can be chosen:
// POLICY 0 plain, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 sc1 nt.  Question: does any policy make the L2 fetch 64 B
// (TCC_EA0_RDREQ_64B) instead of a whole 128-B line for a 64-B row?
template <int POLICY>
__device__ __forceinline__ f32x4 policy_load(const float* p) {
  f32x4 v;
  if (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int POLICY, bool COPY>
__global__ void __launch_bounds__(kBlock)
policy_kernel(const float* __restrict__ tab, int64_t V, int E, int F, const int64_t* __restrict__ idx, uint32_t items,
              float* __restrict__ out, float* __restrict__ sink) {
  constexpr int UNROLL = 2;
  const int lpr = E / 4;
  const int sub = threadIdx.x % lpr;
  const uint32_t rpb = kBlock / lpr;
  const uint32_t per_block = rpb * UNROLL;
  const uint32_t slot = threadIdx.x / lpr;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (uint32_t base = blockIdx.x * per_block; base < items; base += gridDim.x * per_block) {
    const float* src[UNROLL];
    float* dst[UNROLL];
    bool live[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      const uint32_t i = base + slot + k * rpb;
      live[k] = i < items;
      const uint32_t ii = live[k] ? i : 0;
      const uint32_t f = ii % (uint32_t)F;
      src[k] = tab + ((int64_t)f * V + idx[ii]) * E + sub * 4;
      dst[k] = out + (int64_t)ii * E + sub * 4;
    }
    f32x4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) v[k] = policy_load<POLICY>(src[k]);
    // the wait takes the loaded registers as in/out operands: no use of them can be scheduled above it
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]) : : "memory");
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      if (COPY) {
        if (live[k]) *(GLOBAL f32x4*)(dst[k]) = v[k];
      } else {
        acc += v[k];
      }
    }
  }
  if (!COPY && acc.x + acc.y + acc.z + acc.w == -12345.678f) sink[threadIdx.x] = acc.x;
}

// scatter-add of gout rows into grad rows: one dword per lane (16 lanes per 64-B row)
template <bool ATOMIC, typename IdxT>
__global__ void __launch_bounds__(kBlock)
scatter_kernel(float* __restrict__ grad, int64_t V, int E, int F, const IdxT* __restrict__ idx,
               const uint32_t* __restrict__ perm, const uint32_t* __restrict__ field_of, uint32_t items,
               const float* __restrict__ gout) {
  const uint32_t total = items * (uint32_t)E;
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gridDim.x * blockDim.x) {
    const uint32_t i = g / (uint32_t)E, e = g % (uint32_t)E;
    uint32_t f, d;
    if (perm) {
      f = field_of[i];
      d = perm[i];
    } else {
      f = i % (uint32_t)F;
      d = i;
    }
    const int64_t r = (int64_t)idx[i];
    const float v = gout[(int64_t)d * E + e];
    float* p = grad + ((int64_t)f * V + r) * E + e;
    if (ATOMIC) (void)__builtin_amdgcn_global_atomic_fadd_f32((GLOBAL float*)p, v);
    else *(GLOBAL float*)p = v;
  }
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  template <typename Fn>
  float us(Fn fn, int reps) {
    fn();
    fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms * 1e3f / reps;
  }
};

int main(int argc, char** argv) {
  const bool quick = argc > 1 && (!strcmp(argv[1], "quick") || !strcmp(argv[1], "policy"));
  const bool policy_only = argc > 1 && !strcmp(argv[1], "policy");
  const int F = 26, B = 65536;
  const uint32_t items = (uint32_t)F * B;
  Timer T;
  const int reps = 20;
  float* sink;
  CK(hipMalloc(&sink, 4096));
  const int Es[] = {16, 32, 64};
  const int64_t Vs[] = {10000, 100000, 1000000};
  printf("%-34s %3s %8s %8s %9s %9s\n", "variant", "E", "V", "us", "GB/s_alg", "Grows/s");
  for (int E : Es) {
    if (quick && E != 16) continue;
    for (int64_t V : Vs) {
      if (quick && V != 1000000) continue;
      const size_t tab_floats = (size_t)F * V * E;
      float *tab, *out, *grad = nullptr;
      CK(hipMalloc(&tab, tab_floats * 4));
      CK(hipMalloc(&out, (size_t)items * E * 4));
      fill_kernel<<<2048, 256>>>(tab, tab_floats, 17u);
      fill_kernel<<<2048, 256>>>(out, (size_t)items * E, 3u);
      for (int dist = 0; dist < 2; ++dist) {  // 0 uniform, 1 zipf(~1)
        std::mt19937_64 rng(1234 + dist);
        std::vector<int64_t> h_idx(items);
        std::uniform_real_distribution<double> U(0.0, 1.0);
        for (uint32_t i = 0; i < items; ++i) {
          if (dist == 0) h_idx[i] = (int64_t)(rng() % (uint64_t)V);
          else {
            int64_t r = (int64_t)(std::pow((double)V, U(rng)) - 1.0);
            h_idx[i] = std::min<int64_t>(std::max<int64_t>(r, 0), V - 1);
          }
        }
        // sorted per field: list grouped by field, each group sorted by row; perm = original item
        std::vector<uint32_t> h_perm(items), h_field(items);
        std::vector<int64_t> h_sidx(items);
        {
          std::vector<std::pair<int64_t, uint32_t>> tmp(B);
          uint32_t o = 0;
          for (int f = 0; f < F; ++f) {
            for (int b = 0; b < B; ++b) tmp[b] = {h_idx[(size_t)b * F + f], (uint32_t)(b * F + f)};
            std::sort(tmp.begin(), tmp.end());
            for (int b = 0; b < B; ++b, ++o) {
              h_sidx[o] = tmp[b].first;
              h_perm[o] = tmp[b].second;
              h_field[o] = (uint32_t)f;
            }
          }
        }
        // interleaved sorted: position j of every field's sorted list next to each other (keeps 26 streams sweeping)
        std::vector<uint32_t> h_perm2(items), h_field2(items);
        std::vector<int64_t> h_sidx2(items);
        for (int b = 0; b < B; ++b)
          for (int f = 0; f < F; ++f) {
            const size_t s = (size_t)f * B + b, d = (size_t)b * F + f;
            h_sidx2[d] = h_sidx[s];
            h_perm2[d] = h_perm[s];
            h_field2[d] = h_field[s];
          }
        std::vector<int32_t> h_idx32(items);
        for (uint32_t i = 0; i < items; ++i) h_idx32[i] = (int32_t)h_idx[i];
        int64_t *idx, *sidx, *sidx2;
        int32_t* idx32;
        uint32_t *perm, *field, *perm2, *field2;
        CK(hipMalloc(&idx, items * 8)); CK(hipMalloc(&sidx, items * 8)); CK(hipMalloc(&sidx2, items * 8));
        CK(hipMalloc(&idx32, items * 4));
        CK(hipMalloc(&perm, items * 4)); CK(hipMalloc(&field, items * 4));
        CK(hipMalloc(&perm2, items * 4)); CK(hipMalloc(&field2, items * 4));
        CK(hipMemcpy(idx, h_idx.data(), items * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(sidx, h_sidx.data(), items * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(sidx2, h_sidx2.data(), items * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(idx32, h_idx32.data(), items * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(perm, h_perm.data(), items * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(field, h_field.data(), items * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(perm2, h_perm2.data(), items * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(field2, h_field2.data(), items * 4, hipMemcpyHostToDevice));
        const char* dn = dist ? "zipf" : "unif";
        const double row_b = E * 4.0;
        auto report = [&](const char* name, float us, double bytes) {
          char nm[64];
          snprintf(nm, sizeof nm, "%s/%s", name, dn);
          printf("%-34s %3d %8lld %8.1f %9.1f %9.2f\n", nm, E, (long long)V, us, bytes / us / 1e3, items / us / 1e3);
          fflush(stdout);
        };
        const int lpr = E / 4;
        auto grid_for = [&](int unroll, int cap) {
          int64_t g = (items + (kBlock / lpr) * unroll - 1) / ((kBlock / lpr) * unroll);
          return (int)std::min<int64_t>(g, cap);
        };
        const double copy_b = items * (2 * row_b + 8), read_b = items * (row_b + 8), write_b = items * row_b;
#define POL(P) \
  report("policy" #P " read", T.us([&] { policy_kernel<P, false><<<8192, kBlock>>>(tab, V, E, F, idx, items, out, sink); }, reps), read_b); \
  report("policy" #P " copy", T.us([&] { policy_kernel<P, true><<<8192, kBlock>>>(tab, V, E, F, idx, items, out, sink); }, reps), copy_b);
#define RUN(UNR, MODE, PERMF, IDX, PERM, FIELD, CAP) \
  T.us([&] { rows_kernel<UNR, MODE, PERMF><<<grid_for(UNR, CAP), kBlock>>>(tab, V, E, F, IDX, PERM, FIELD, items, out, sink); }, reps)
        report("copy u4 g4096", RUN(4, 0, false, idx, nullptr, nullptr, 4096), copy_b);
        if (policy_only) {
          report("copy u2 g8192", RUN(2, 0, false, idx, nullptr, nullptr, 8192), copy_b);
          report("read u2 g8192", RUN(2, 1, false, idx, nullptr, nullptr, 8192), read_b);
          POL(0) POL(1) POL(2) POL(3) POL(4) POL(5)
          CK(hipFree(idx)); CK(hipFree(sidx)); CK(hipFree(sidx2)); CK(hipFree(idx32));
          CK(hipFree(perm)); CK(hipFree(field)); CK(hipFree(perm2)); CK(hipFree(field2));
          continue;
        }
        if (!quick || true) {
          report("copy u1 g8192", RUN(1, 0, false, idx, nullptr, nullptr, 8192), copy_b);
          report("copy u2 g8192", RUN(2, 0, false, idx, nullptr, nullptr, 8192), copy_b);
          report("copy u8 g2048", RUN(8, 0, false, idx, nullptr, nullptr, 2048), copy_b);
          report("copy u4 g1024", RUN(4, 0, false, idx, nullptr, nullptr, 1024), copy_b);
          report("copy u4 g65536", RUN(4, 0, false, idx, nullptr, nullptr, 1 << 16), copy_b);
          report("copy u4 idx32", RUN(4, 0, false, idx32, nullptr, nullptr, 4096), copy_b - items * 4.0);
        }
        report("read u4 g4096", RUN(4, 1, false, idx, nullptr, nullptr, 4096), read_b);
        report("read u8 g4096", RUN(8, 1, false, idx, nullptr, nullptr, 4096), read_b);
        report("read u8 g65536", RUN(8, 1, false, idx, nullptr, nullptr, 1 << 16), read_b);
        if (dist == 0) report("write u4 g4096", RUN(4, 2, false, idx, nullptr, nullptr, 4096), write_b);
        report("read sorted-by-field u4", RUN(4, 1, true, sidx, perm, field, 4096), read_b + items * 8.0);
        report("read sorted-interleaved u4", RUN(4, 1, true, sidx2, perm2, field2, 4096), read_b + items * 8.0);
        report("copy sorted-by-field u4", RUN(4, 0, true, sidx, perm, field, 4096), copy_b + items * 8.0);
        report("copy sorted-interleaved u4", RUN(4, 0, true, sidx2, perm2, field2, 4096), copy_b + items * 8.0);
        // scatter side (the backward): gout row -> grad row
        if (E == 16 || !quick) {
          CK(hipMalloc(&grad, tab_floats * 4));
          CK(hipMemset(grad, 0, tab_floats * 4));
          const double sc_b = items * (3 * row_b + 8);
          const int sg = 2048;
          report("scatter atomic", T.us([&] { scatter_kernel<true><<<sg, kBlock>>>(grad, V, E, F, idx, nullptr, nullptr, items, out); }, reps), sc_b);
          report("scatter atomic g8192", T.us([&] { scatter_kernel<true><<<8192, kBlock>>>(grad, V, E, F, idx, nullptr, nullptr, items, out); }, reps), sc_b);
          report("scatter store(no add)", T.us([&] { scatter_kernel<false><<<sg, kBlock>>>(grad, V, E, F, idx, nullptr, nullptr, items, out); }, reps), sc_b - items * row_b);
          report("scatter atomic sorted-by-field", T.us([&] { scatter_kernel<true><<<sg, kBlock>>>(grad, V, E, F, sidx, perm, field, items, out); }, reps), sc_b + items * 8.0);
          report("scatter atomic sorted-interl", T.us([&] { scatter_kernel<true><<<sg, kBlock>>>(grad, V, E, F, sidx2, perm2, field2, items, out); }, reps), sc_b + items * 8.0);
          CK(hipFree(grad));
        }
        CK(hipFree(idx)); CK(hipFree(sidx)); CK(hipFree(sidx2)); CK(hipFree(idx32));
        CK(hipFree(perm)); CK(hipFree(field)); CK(hipFree(perm2)); CK(hipFree(field2));
      }
      CK(hipFree(tab));
      CK(hipFree(out));
    }
  }
  return 0;
}
