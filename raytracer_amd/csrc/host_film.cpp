// Film helpers on the host side of libmirt: the tonemap threshold table and the BMP writer.
//
// Tonemap (reference util/image.cuh:36-48): v = int(255 * max(0, min(1, powf(c * sqrt(2), 1/2.2)))) per channel.
// powf is a libm function and device libm differs from the host's in the last ulp, so instead of evaluating powf on the
// GPU the library tabulates, once, with the HOST libm the reference itself would use, the 255 thresholds
// T[k] = smallest float c >= 0 with v(c) >= k.  v is a monotone step function of c, so v(c) = #{k : c >= T[k]} -- a
// 8-step binary search on the device that is exact by construction (and cheaper than powf).
#include "host_film.hpp"

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <vector>

namespace mirt {
namespace {

inline float maxf(float a, float b) { return (a < b) ? b : a; }   // std::max / std::min, util/misc.cuh:28-29
inline float minf(float a, float b) { return (b < a) ? b : a; }

inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

}  // namespace

int tonemap_channel_direct(float c) {
  const float gamma = 2.2f;
  const float level = 1.0f;
  const float one_over_gamma = 1.0f / gamma;
  const float exposure = (float) std::sqrt(std::pow(2.0, (double) level));
  return (int) (255 * maxf(0.f, minf(1.f, powf(c * exposure, one_over_gamma))));
}

bool build_tonemap_table(float table[256]) {
  // non-negative floats are ordered like their bit patterns: binary search on the bits in [+0, +inf]
  bool monotone = true;
  table[0] = 0.f;
  for (int k = 1; k <= 255; k++) {
    uint32_t lo = 0u, hi = 0x7f800000u;            // invariant: v(lo) < k <= v(hi); v(+0) = 0, v(+inf) = 255
    while (hi - lo > 1) {
      uint32_t mid = lo + (hi - lo) / 2;
      if (tonemap_channel_direct(u2f(mid)) >= k) hi = mid; else lo = mid;
    }
    table[k] = u2f(hi);
    for (uint32_t d = 1; d <= 16; d++) {           // local monotonicity check around the threshold
      if (hi >= d && tonemap_channel_direct(u2f(hi - d)) >= k) monotone = false;
      if (hi + d <= 0x7f800000u && tonemap_channel_direct(u2f(hi + d)) < k) monotone = false;
    }
  }
  return monotone;
}

int tonemap_channel_table(const float table[256], float c) {
  if (c != c) return 255;                          // powf(NaN) -> min(1, NaN) = 1
  if (c < 0.f) return 255;                         // powf(negative, 1/2.2) = NaN
  int lo = 0, hi = 256;                            // largest k with c >= table[k]; table[0] = 0
  while (hi - lo > 1) {
    int mid = (lo + hi) / 2;
    if (c >= table[mid]) lo = mid; else hi = mid;
  }
  return lo;
}

// 24-bit BMP exactly as bitmap_image::save_image writes it (util/bitmap_image.cuh:385-430, :1142-1192):
// 14-byte file header (type 19778, off_bits 54) + 40-byte info header (planes 1, 24 bpp, compression 0,
// size_image = row_stride * h), rows bottom-up, BGR, each padded to a multiple of 4 bytes.
int write_bmp(const char *path, const int32_t *rgb, uint32_t w, uint32_t h) {
  const uint32_t row = w * 3;
  const uint32_t pad = (4 - ((3 * w) % 4)) % 4;
  const uint32_t size_image = (((w * 3) + 3) & 0x0000FFFC) * h;
  std::vector<unsigned char> buf;
  buf.reserve(54 + (size_t) (row + pad) * h);
  auto put16 = [&](uint16_t v) { buf.push_back(v & 0xff); buf.push_back(v >> 8); };
  auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) buf.push_back((v >> (8 * i)) & 0xff); };
  put16(19778);
  put32(54 + size_image);
  put16(0);
  put16(0);
  put32(54);
  put32(40);
  put32(w);
  put32(h);
  put16(1);
  put16(24);
  put32(0);
  put32(size_image);
  put32(0);
  put32(0);
  put32(0);
  put32(0);
  for (uint32_t i = 0; i < h; i++) {
    const int32_t *src = rgb + (size_t) (h - i - 1) * w * 3;
    for (uint32_t x = 0; x < w; x++) {
      buf.push_back((unsigned char) src[3 * x + 2]);
      buf.push_back((unsigned char) src[3 * x + 1]);
      buf.push_back((unsigned char) src[3 * x + 0]);
    }
    for (uint32_t p = 0; p < pad; p++) buf.push_back(0);
  }
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  size_t n = fwrite(buf.data(), 1, buf.size(), f);
  fclose(f);
  return n == buf.size() ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------ film widening
// The film crosses PCIe as one word per pixel and is widened on the host to the reference's Vector3i records.  The scalar
// loop writes 12 bytes per pixel through the cache (read-for-ownership of every line of the caller's 25 MB buffer): one
// thread widens a 1/8 piece of a 1080p film in ~0.5 ms, which sat between the last copy and the next frame.  With SSSE3 four
// pixels become three byte shuffles and three non-temporal 16-byte stores.
#if defined(__x86_64__)
__attribute__((target("ssse3"))) static void widen_ssse3(const uint32_t *packed, size_t n, int32_t *rgb) {
  const __m128i s0 = _mm_setr_epi8(0, -1, -1, -1, 1, -1, -1, -1, 2, -1, -1, -1, 4, -1, -1, -1);      // r0 g0 b0 r1
  const __m128i s1 = _mm_setr_epi8(5, -1, -1, -1, 6, -1, -1, -1, 8, -1, -1, -1, 9, -1, -1, -1);      // g1 b1 r2 g2
  const __m128i s2 = _mm_setr_epi8(10, -1, -1, -1, 12, -1, -1, -1, 13, -1, -1, -1, 14, -1, -1, -1);  // b2 r3 g3 b3
  size_t i = 0;
  for (; i + 4 <= n; i += 4) {
    const __m128i v = _mm_loadu_si128((const __m128i *) (packed + i));
    __m128i *o = (__m128i *) (rgb + 3 * i);
    _mm_stream_si128(o, _mm_shuffle_epi8(v, s0));
    _mm_stream_si128(o + 1, _mm_shuffle_epi8(v, s1));
    _mm_stream_si128(o + 2, _mm_shuffle_epi8(v, s2));
  }
  _mm_sfence();
  for (; i < n; i++) {
    const uint32_t v = packed[i];
    rgb[3 * i] = (int32_t) (v & 0xFFu); rgb[3 * i + 1] = (int32_t) ((v >> 8) & 0xFFu); rgb[3 * i + 2] = (int32_t) ((v >> 16) & 0xFFu);
  }
}
#endif

void widen_film(const uint32_t *packed, size_t n, int32_t *rgb) {
  size_t i = 0;
#if defined(__x86_64__)
  if (__builtin_cpu_supports("ssse3")) {
    // scalar until the output is 16-byte aligned (a pixel is 12 bytes: at most three pixels, if the buffer is int-aligned)
    while (i < n && (((uintptr_t) (rgb + 3 * i)) & 15u) != 0 && i < 4) {
      const uint32_t v = packed[i];
      rgb[3 * i] = (int32_t) (v & 0xFFu); rgb[3 * i + 1] = (int32_t) ((v >> 8) & 0xFFu); rgb[3 * i + 2] = (int32_t) ((v >> 16) & 0xFFu);
      i++;
    }
    if ((((uintptr_t) (rgb + 3 * i)) & 15u) == 0) {
      widen_ssse3(packed + i, n - i, rgb + 3 * i);
      return;
    }
  }
#endif
  for (; i < n; i++) {
    const uint32_t v = packed[i];
    rgb[3 * i] = (int32_t) (v & 0xFFu); rgb[3 * i + 1] = (int32_t) ((v >> 8) & 0xFFu); rgb[3 * i + 2] = (int32_t) ((v >> 16) & 0xFFu);
  }
}

}  // namespace mirt
