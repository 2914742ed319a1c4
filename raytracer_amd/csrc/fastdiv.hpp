// Exact unsigned division by a run-time constant without the ~25-instruction udiv sequence: k_logic divides by the number of
// pixel slots, by the tiles per image row and by the pool's group size several times per slot and iteration, and those
// sequences were a fifth of its vector instructions.  Round-up method (Granlund & Montgomery 1994): for 0 <= n < 2^31 and
// d >= 1, with l = ceil(log2 d) and m = floor(2^(31+l) / d) + 1 (< 2^32), floor(n / d) = (n * m) >> (31 + l), because
// 2^(31+l) < m * d <= 2^(31+l) + 2^l.  The host checks the 2^31 bound (capi.hip); tests/native/host_pool_check.cpp checks
// the identity against `/` on edge cases and random pairs.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MIRT_HD __host__ __device__
#else
#define MIRT_HD
#endif

namespace mirt {

struct FastDiv {
  uint32_t m, sh;
};

inline FastDiv make_fastdiv(uint32_t d) {
  if (d == 0) d = 1;
  uint32_t l = 0;
  while ((1ull << l) < d) l++;
  FastDiv f;
  f.m = (uint32_t) (((1ull << (31 + l)) / d) + 1);
  f.sh = 31 + l;
  return f;
}

MIRT_HD inline uint32_t fdiv(uint32_t n, FastDiv f) { return (uint32_t) (((uint64_t) n * f.m) >> f.sh); }   // n < 2^31

}  // namespace mirt
