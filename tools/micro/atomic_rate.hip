// Same-address returning atomicAdd throughput (one lane per wave), for sizing queue-reservation schemes.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(uint32_t *ctr, uint32_t *sink, int reps, int naddr) {
  uint32_t acc = 0;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < reps; i++) acc += atomicAdd(ctr + 32 * ((wave + i) % naddr), 1u);
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
int main() {
  uint32_t *ctr, *sink;
  hipMalloc(&ctr, 4096 * 128); hipMalloc(&sink, 4);
  hipMemset(ctr, 0, 4096 * 128);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int naddr : {1, 2, 4, 16, 256}) {
    for (int blocks : {256, 2048, 8192}) {
      const int reps = 64;
      k<<<blocks, 256>>>(ctr, sink, reps, naddr);
      hipEventRecord(a);
      k<<<blocks, 256>>>(ctr, sink, reps, naddr);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      double n = (double) blocks * 4 * reps;
      printf("naddr %4d blocks %5d: %.3f ms, %.1f M atomics/s, %.2f ns each\n", naddr, blocks, ms, n / ms / 1e3, ms * 1e6 / n);
    }
  }
  return 0;
}
