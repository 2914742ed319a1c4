// Film helpers on the host side of libmirt (see host_film.cpp).
#pragma once
#include <cstddef>
#include <cstdint>

namespace mirt {
int tonemap_channel_direct(float c);                       // util/image.cuh:36-48 evaluated with the host libm
bool build_tonemap_table(float table[256]);                // returns false if the local monotonicity check failed
int tonemap_channel_table(const float table[256], float c);
int write_bmp(const char *path, const int32_t *rgb, uint32_t w, uint32_t h);
// packed film words (r | g << 8 | b << 16) -> the reference's three ints per pixel (SampleBuffer::data, util/image.cuh:9-48)
void widen_film(const uint32_t *packed, size_t n, int32_t *rgb);
}  // namespace mirt
