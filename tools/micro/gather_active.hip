// Does a divergent 16-byte gather cost per INSTRUCTION or per ACTIVE LANE?  Every wave runs REPS dependent steps; in each
// step only A of its 64 lanes (the first A, or every (64/A)-th) read NL float4 of a pseudo-random 128-byte record.  Also: the
// same reads from a 73-record table in LDS (the top of a tree cached per workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template<int NL, bool LDS>
__global__ void __launch_bounds__(256) k(const float4 *tab, uint32_t nrec, int reps, int active, int strided, float *sink) {
  __shared__ float4 top[73 * 6];
  if (LDS) {
    for (int i = threadIdx.x; i < 73 * 6; i += 256) top[i] = tab[i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const bool on = strided ? (lane % (64 / active) == 0) : (lane < active);
  uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0;
  for (int i = 0; i < reps; i++) {
    x = x * 1664525u + 1013904223u;
    if (on) {
      if (LDS) {
        const float4 *p = top + ((x >> 8) % 73u) * 6;
#pragma unroll
        for (int j = 0; j < NL; j++) { float4 v = p[j]; acc += v.x + v.w; }
      } else {
        const float4 *p = tab + (size_t) ((x >> 8) % nrec) * 8;
#pragma unroll
        for (int j = 0; j < NL; j++) { float4 v = p[j]; acc += v.x + v.w; }
      }
    }
    x += (uint32_t) (acc > 1e30f);
  }
  if (acc == 123.456f) sink[0] = acc;
}
template<int NL, bool LDS> void run(const float4 *tab, uint32_t nrec, float *sink, int active, int strided) {
  hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
  const int reps = 200, blocks = 256 * 6;
  k<NL, LDS><<<blocks, 256>>>(tab, nrec, reps, active, strided, sink);
  (void) hipEventRecord(a);
  k<NL, LDS><<<blocks, 256>>>(tab, nrec, reps, active, strided, sink);
  (void) hipEventRecord(b); (void) hipEventSynchronize(b);
  float ms; (void) hipEventElapsedTime(&ms, a, b);
  const double wave_steps = (double) blocks * 4 * reps;
  printf("%s NL=%d active=%2d %s: %.3f ms  %.2f G wave-steps/s  %.1f G lane-loads/s\n", LDS ? "LDS   " : "global", NL, active,
         strided ? "strided" : "first  ", ms, wave_steps / ms / 1e6, wave_steps * active * NL / ms / 1e6);
}
int main() {
  float *sink; (void) hipMalloc(&sink, 4);
  const size_t mb = 20;
  uint32_t nrec = (uint32_t) (mb * 1024 * 1024 / 128);
  float4 *tab; (void) hipMalloc(&tab, (size_t) nrec * 128); (void) hipMemset(tab, 0, (size_t) nrec * 128);
  for (int strided = 0; strided < 2; strided++)
    for (int a : {64, 32, 16, 8, 4, 1}) run<6, false>(tab, nrec, sink, a, strided);
  for (int a : {64, 32, 16}) run<3, false>(tab, nrec, sink, a, 1);
  for (int a : {64, 32, 16}) run<6, true>(tab, nrec, sink, a, 1);
  return 0;
}
