// round 3 microbenchmark: how long does a wave wait for a coalesced 8-byte-per-lane load of a fresh line when every wave
// of the chip issues one at the same moment (the id loads of the NeuralCF kernels)?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void __launch_bounds__(256) probe(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t m, int iters,
                                             int gap, long long* stamps, int64_t* sink) {
  const int lane = threadIdx.x & 63, n = lane & 15;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
  int64_t acc = 0;
  for (int k = 0; k < iters; ++k) {
    int64_t row = (wave0 + k * nwaves) * 16 + n;
    row = row < m ? row : m - 1;
    const long long t0 = __builtin_readcyclecounter();
    const int64_t u = a[row], i = b[row];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += u + i;
    asm volatile("" : "+v"(acc));
    const long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 7 && threadIdx.x == 0) stamps[k] = t1 - t0;
    for (int s = 0; s < gap; ++s) asm volatile("s_sleep 8");
  }
  if (acc == 0x7fffffffffffll) sink[0] = acc;
}

__global__ void __launch_bounds__(256) dirty(float4* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = make_float4(1, 2, 3, 4);
}

int main() {
  const int64_t m = 65536 * 4;
  int64_t *a, *b, *sink;
  long long* st;
  float4* big;
  hipMalloc(&a, m * 8); hipMalloc(&b, m * 8); hipMalloc(&sink, 8); hipMalloc(&st, 64 * 8);
  hipMalloc(&big, (size_t)64 << 20);
  std::vector<int64_t> h(m);
  for (int64_t i = 0; i < m; ++i) h[i] = i % 943;
  hipMemcpy(a, h.data(), m * 8, hipMemcpyHostToDevice);
  hipMemcpy(b, h.data(), m * 8, hipMemcpyHostToDevice);
  long long out[64];
  for (int grid : {256, 1024})
    for (int gap : {0, 40})
      for (int pre : {0, 1}) {
        for (int rep = 0; rep < 3; ++rep) {
          if (pre) hipLaunchKernelGGL(dirty, dim3(2048), dim3(256), 0, 0, big, (int64_t)(64 << 20) / 16);
          hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, a, b, m, 6, gap, st, sink);
        }
        hipDeviceSynchronize();
        hipMemcpy(out, st, 6 * 8, hipMemcpyDeviceToHost);
        printf("grid %4d gap %2d dirty-before %d: wait cycles per iteration:", grid, gap, pre);
        for (int k = 0; k < 6; ++k) printf(" %lld", out[k]);
        printf("\n");
      }
  return 0;
}
