// Film helpers on the host side of libmirt (see host_film.cpp).
#pragma once
#include <cstdint>

namespace mirt {
int tonemap_channel_direct(float c);                       // util/image.cuh:36-48 evaluated with the host libm
bool build_tonemap_table(float table[256]);                // returns false if the local monotonicity check failed
int tonemap_channel_table(const float table[256], float c);
int write_bmp(const char *path, const int32_t *rgb, uint32_t w, uint32_t h);
}  // namespace mirt
