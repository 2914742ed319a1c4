// Divergent gather throughput: every lane reads NL float4 of its own pseudo-random 128-byte record, REPS times.
// Tells whether the vector L1 is bound by lane requests, by lines touched or by bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template<int NL>
__global__ void __launch_bounds__(256) k(const float4 *tab, uint32_t nrec, int reps, float *sink) {
  uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0;
  for (int i = 0; i < reps; i++) {
    x = x * 1664525u + 1013904223u;
    uint32_t rec = (x >> 8) % nrec;
    const float4 *p = tab + (size_t) rec * 8;
#pragma unroll
    for (int j = 0; j < NL; j++) { float4 v = p[j]; acc += v.x + v.w; }
    x += (uint32_t) (acc > 1e30f);   // dependent, like a traversal
  }
  if (acc == 123.456f) sink[0] = acc;
}
template<int NL> void run(const float4 *tab, uint32_t nrec, float *sink, const char *what) {
  hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
  const int reps = 200, blocks = 256 * 6;
  k<NL><<<blocks, 256>>>(tab, nrec, reps, sink);
  (void) hipEventRecord(a);
  k<NL><<<blocks, 256>>>(tab, nrec, reps, sink);
  (void) hipEventRecord(b); (void) hipEventSynchronize(b);
  float ms; (void) hipEventElapsedTime(&ms, a, b);
  double recs = (double) blocks * 256 * reps;
  printf("%s NL=%d: %.3f ms  %.2f G records/s  %.2f G lane-loads/s  %.2f TB/s\n", what, NL, ms, recs / ms / 1e6, recs * NL / ms / 1e6, recs * NL * 16 / ms / 1e9);
}
int main() {
  float *sink; (void) hipMalloc(&sink, 4);
  for (size_t mb : {2, 20, 200}) {
    uint32_t nrec = (uint32_t) (mb * 1024 * 1024 / 128);
    float4 *tab; (void) hipMalloc(&tab, (size_t) nrec * 128); (void) hipMemset(tab, 0, (size_t) nrec * 128);
    char what[64]; snprintf(what, sizeof what, "table %zu MB", mb);
    run<1>(tab, nrec, sink, what); run<2>(tab, nrec, sink, what); run<3>(tab, nrec, sink, what); run<4>(tab, nrec, sink, what); run<7>(tab, nrec, sink, what);
    (void) hipFree(tab);
  }
  return 0;
}
