// CPU check of two host-side helpers of libmirt that need no GPU (tests/test_host.py compiles this against
// raytracer_amd/csrc/host_bvh.cpp and host_film.cpp): WorkerPool runs every job exactly once, run after run, with more and
// with fewer jobs than threads; widen_film equals the scalar definition for every length / alignment.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "fastdiv.hpp"
#include "host_bvh.hpp"
#include "host_film.hpp"

int main() {
  for (unsigned workers : {0u, 1u, 3u, 15u}) {
    mirt::WorkerPool pool(workers);
    for (int round = 0; round < 300; round++) {
      const size_t n = (size_t) (round % 37);
      std::vector<std::atomic<int>> hits(n);
      for (auto &h : hits) h.store(0);
      pool.run(n, [&](size_t j) { hits[j].fetch_add(1); });
      for (size_t j = 0; j < n; j++)
        if (hits[j].load() != 1) { printf("pool(%u) round %d: job %zu ran %d times\n", workers, round, j, hits[j].load()); return 1; }
    }
  }
  for (size_t n : {0ul, 1ul, 3ul, 4ul, 5ul, 7ul, 1000ul, 259200ul})
    for (int off = 0; off < 4; off++) {
      std::vector<uint32_t> in(n + 8);
      for (size_t i = 0; i < n + 8; i++) in[i] = (uint32_t) (i * 2654435761u) & 0xFFFFFFu;
      std::vector<int32_t> out(3 * n + 16, -7), want(3 * n + 16, -7);
      for (size_t i = 0; i < n; i++) {
        const uint32_t v = in[i + 1];
        want[off + 3 * i] = v & 255; want[off + 3 * i + 1] = (v >> 8) & 255; want[off + 3 * i + 2] = (v >> 16) & 255;
      }
      mirt::widen_film(in.data() + 1, n, out.data() + off);
      if (out != want) { printf("widen_film mismatch n=%zu off=%d\n", n, off); return 1; }
    }
  // fastdiv.hpp: (n * m) >> sh equals n / d for every dividend below 2^31
  {
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    const uint32_t ds[] = {1, 2, 3, 5, 7, 63, 64, 65, 240, 255, 256, 257, 1000, 32400, 65535, 65536, 65537, 2073600, 8294400, 16777215,
                           16777216, 16777217, 0x3FFFFFFFu, 0x40000000u, 0x40000001u, 0x7FFFFFFEu, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFFu};
    std::vector<uint32_t> dl(ds, ds + sizeof ds / sizeof ds[0]);
    for (int i = 0; i < 3000; i++) dl.push_back((uint32_t) (rnd() >> (32 + rnd() % 31)) | 1u);
    for (uint32_t d : dl) {
      const mirt::FastDiv f = mirt::make_fastdiv(d);
      std::vector<uint32_t> ns = {0u, 1u, d - 1, d, d + 1, 2 * d - 1, 2 * d, 0x7FFFFFFFu, 0x7FFFFFFEu, 0x40000000u};
      for (uint64_t k = 1; k <= 40; k++) { ns.push_back((uint32_t) (k * d - 1)); ns.push_back((uint32_t) (k * d)); }
      for (int i = 0; i < 300; i++) ns.push_back((uint32_t) rnd());
      const uint64_t top = (0x7FFFFFFFull / d) * d;                     // the largest multiple of d below 2^31 and its neighbours
      ns.push_back((uint32_t) top); ns.push_back((uint32_t) (top ? top - 1 : 0));
      for (uint32_t n : ns) {
        n &= 0x7FFFFFFFu;
        if (mirt::fdiv(n, f) != n / d) { printf("fastdiv: %u / %u = %u, got %u\n", n, d, n / d, mirt::fdiv(n, f)); return 1; }
      }
    }
  }
  printf("ok\n");
  return 0;
}
