#include "host_bvh.hpp"
#include <cstdio>
#include <random>
using namespace mirt;
int main() {
  std::mt19937 g(7);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  for (int round = 0; round < 4; round++) {
    const uint32_t n = round == 0 ? 3 : (round == 1 ? 1000 : (round == 2 ? 60000 : 200000));
    std::vector<float> pos(9 * (size_t) n);
    std::vector<uint8_t> type(n, 2);
    for (uint32_t i = 0; i < n; i++) {
      float c[3] = {u(g) * 3, u(g) * 3, u(g) * 3};
      if (round == 3 && i % 3 == 0) {  // sphere
        type[i] = 1;
        pos[9 * i] = c[0]; pos[9 * i + 1] = c[1]; pos[9 * i + 2] = c[2]; pos[9 * i + 3] = 0.01f + 0.05f * std::fabs(u(g));
      } else {
        for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) pos[9 * i + 3 * v + k] = c[k] + 0.05f * u(g);
      }
      if (round == 2 && i % 1000 == 0 && i) for (int k = 0; k < 9; k++) pos[9 * i + k] = pos[9 * (i - 1) + k];  // duplicates
    }
    HostBVH b;
    build_reference_bvh2(pos.data(), type.data(), n, 4, b);
    collapse_to_wide(b);
    std::vector<float> flat(9 * (size_t) n);
    std::vector<uint8_t> tflat(n);
    for (uint32_t i = 0; i < n; i++) { for (int k = 0; k < 9; k++) flat[9 * i + k] = pos[9 * b.flat_to_orig[i] + k]; tflat[i] = type[b.flat_to_orig[i]]; }
    build_fast_tree(flat.data(), tflat.data(), n, b);
    std::printf("n %u: nodes2 %zu nodes4 %zu nodes8 %zu depth %u/%u/%u\n", n, b.nodes2.size(), b.nodes4.size(), b.nodes8q.size() / 24, b.depth2, b.depth4, b.depth8);
  }
  return 0;
}
