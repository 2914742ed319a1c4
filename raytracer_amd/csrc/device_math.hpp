// Device-side vector math with the REFERENCE's operation order and promotions (linalg/Vector3f.cuh, Matrix3f.cuh),
// so that, compiled with -ffp-contract=off and correctly rounded divide/sqrt, the gfx950 code rounds exactly as the
// reference's g++ build does.  Notes that make the float forms below exact restatements of the reference's double forms:
//   * `float rc = 1.0 / c` (double divide, then narrowing) == 1.0f / c in correctly rounded float division: rounding a
//     53-bit quotient to 24 bits is innocuous for division and sqrt (p >= 2q + 2);
//   * `sqrt(x)` on a float argument (double sqrt, narrowed) == sqrtf(x) for the same reason.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mirt {

#define MIRT_PI (3.14159265358979323f)     // util/misc.cuh:16
#define MIRT_EPS (0.00001f)                // util/misc.cuh:17
#define MIRT_INF (__builtin_inff())

struct V3 {
  float x, y, z;
};
struct M3 {
  V3 c0, c1, c2;                           // columns (Matrix3f.cuh:165-171)
};

// util/misc.cuh:28-29 on the g++ path (the oracle): std::max / std::min.  Argument order matters with NaN.
__device__ __forceinline__ float smax(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float smin(float a, float b) { return (b < a) ? b : a; }

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 vmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 vdivv(V3 a, V3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ V3 vneg(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 vscale(V3 a, float c) { return mk(a.x * c, a.y * c, a.z * c); }      // v * c  (:101)
__device__ __forceinline__ V3 vlscale(float c, V3 a) { return mk(c * a.x, c * a.y, c * a.z); }     // c * v  (:180)
__device__ __forceinline__ V3 vdiv(V3 a, float c) {                                                // v / c  (:106-109)
  const float rc = 1.0f / c;
  return mk(rc * a.x, rc * a.y, rc * a.z);
}
__device__ __forceinline__ float vdot(V3 a, V3 b) { return b.x * a.x + b.y * a.y + b.z * a.z; }    // a.dot(b) (:50)
__device__ __forceinline__ V3 vcross(V3 a, V3 b) {                                                 // a.cross(b) (:55)
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float vnorm(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }  // :147
__device__ __forceinline__ V3 vunit(V3 a) {                                                        // :161-164
  const float rn = 1.0f / vnorm(a);
  return vscale(a, rn);
}
__device__ __forceinline__ V3 vrcp(V3 a) { return mk(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }        // :140
__device__ __forceinline__ V3 vsplat(float c) { return mk(c, c, c); }
__device__ __forceinline__ float villum(V3 a) { return 0.2126f * a.x + 0.7152f * a.y + 0.0722f * a.z; }   // :173

__device__ __forceinline__ V3 mmulv(const M3 &m, V3 x) {                                           // Matrix3f.cuh:243
  return vadd(vadd(vlscale(x.x, m.c0), vlscale(x.y, m.c1)), vlscale(x.z, m.c2));
}
__device__ __forceinline__ M3 mtranspose(const M3 &a) {                                            // Matrix3f.cuh:130
  M3 b;
  b.c0 = mk(a.c0.x, a.c1.x, a.c2.x);
  b.c1 = mk(a.c0.y, a.c1.y, a.c2.y);
  b.c2 = mk(a.c0.z, a.c1.z, a.c2.z);
  return b;
}

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// ------------------------------------------------------------------------------------------------ RNG
// mode 0: the reference's xorshift32 (util/sampler.cuh:8-15).  mode 1: Philox2x32-10 keyed by (pixel, seed), counter
// (draw index / 2, sample index): any sample of any pixel can be generated on any GPU.
struct Rng {
  uint32_t mode, state, key, sample;
};

__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t ctr_key(uint32_t pixel, uint32_t seed) {
  return mix32(pixel ^ mix32(seed + 0x9E3779B9U));
}
__device__ __forceinline__ uint32_t philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key, bool second) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi = __umulhi(0xD256D193U, c0), lo = 0xD256D193U * c0;
    c0 = hi ^ key ^ c1;
    c1 = lo;
    key += 0x9E3779B9U;
  }
  return second ? c1 : c0;
}
__device__ __forceinline__ float rng_next(Rng &r) {
  uint32_t x;
  if (r.mode == 0) {
    x = r.state;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    r.state = x;
  } else {
    uint32_t n = r.state++;
    x = philox2x32_10(n >> 1, r.sample, r.key, n & 1);
  }
  return (float) x * 2.3283064365386963e-10f;      // float(x) / float(UINT_MAX): the divisor rounds to 2^32
}

}  // namespace mirt
