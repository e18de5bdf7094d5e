// Same-address atomic throughput on gfx950 (run by hand on the GPU box): how fast can many waves bump one counter?
// build: hipcc -O3 --offload-arch=gfx950 atomic_rate.hip -o atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: add with return, 1 address | 1: add no return | 2: umin no return | 3: 3 adds with return, one line | 4: 3 adds with return,
// three lines | 5: 3 umin + 3 umax no return, one line | 6: add with return, address sharded by blockIdx & 7 | 7: same, & 63
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* c, uint32_t* sink) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t r = 0;
  if (lane == 0u) {
    const uint32_t v = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (MODE == 0) r = atomicAdd(c, 1u);
    if (MODE == 1) atomicAdd(c, 1u);
    if (MODE == 2) atomicMin(c, v);
    if (MODE == 3) { r = atomicAdd(c, 1u); r += atomicAdd(c + 1, 1u); r += atomicAdd(c + 2, 1u); }
    if (MODE == 4) { r = atomicAdd(c, 1u); r += atomicAdd(c + 64, 1u); r += atomicAdd(c + 128, 1u); }
    if (MODE == 5) { atomicMin(c, v); atomicMin(c + 1, v); atomicMin(c + 2, v); atomicMax(c + 3, v); atomicMax(c + 4, v); atomicMax(c + 5, v); }
    if (MODE == 6) r = atomicAdd(c + 64u * (blockIdx.x & 7u), 1u);
    if (MODE == 7) r = atomicAdd(c + 64u * (blockIdx.x & 63u), 1u);
  }
  if (r == 0xfffffff0u) sink[0] = r;
}

template <int MODE> int run(const char* what, uint32_t* c, uint32_t* sink, int atomics_per_wave) {
  const int blocks = 16384;  // 65536 waves
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipMemset(c, 0, 64 * 64 * 4));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, c, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, c, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1000.0 / 5.0;
  printf("%-58s %9.1f us / launch  %7.2f ns / atomic\n", what, us, us * 1000.0 / (blocks * 4.0 * atomics_per_wave));
  return 0;
}

int main() {
  uint32_t *c, *sink;
  CK(hipMalloc(&c, 64 * 64 * 4)); CK(hipMalloc(&sink, 64));
  run<0>("add, returns, one address", c, sink, 1);
  run<1>("add, no return, one address", c, sink, 1);
  run<2>("umin, no return, one address", c, sink, 1);
  run<3>("3 adds with return, one line", c, sink, 3);
  run<4>("3 adds with return, three lines", c, sink, 3);
  run<5>("3 umin + 3 umax, no return, one line (k_flatten)", c, sink, 6);
  run<6>("add, returns, 8 addresses by blockIdx & 7", c, sink, 1);
  run<7>("add, returns, 64 addresses by blockIdx & 63", c, sink, 1);
  return 0;
}
