// Dumps what Generator::from_obj (include/raytracer_amd/pathtracer.hpp; reference: geometry/generator.cuh:98-180) makes of
// an OBJ/MTL pair, one line per triangle: 18 floats (p1 p2 p3 n1 n2 n3) then the BSDF record (type + 8 floats).
// Host only: the scene-ingestion tests run it without a GPU.
//   obj_info file.obj [sx sy sz tx ty tz]
#include <cstdio>
#include <cstdlib>
#include <string>

#include "raytracer_amd/pathtracer.hpp"

// obj_info --bin out.bin file.obj [sx sy sz tx ty tz]: the same as one binary file (for meshes of 10^5..10^6 triangles):
// uint32 n, then n x 18 float32 (p1 p2 p3 n1 n2 n3), then n x 9 uint32/float32 words (BSDF type + 8 floats).
static int dump_binary(const char *out, const std::vector<Primitive *> &prims) {
  FILE *f = std::fopen(out, "wb");
  if (!f) return 1;
  const uint32_t n = (uint32_t) prims.size();
  std::fwrite(&n, 4, 1, f);
  for (const Primitive *p: prims) {
    const Vector3f *v[6] = {&p->triangle.p1, &p->triangle.p2, &p->triangle.p3, &p->triangle.n1, &p->triangle.n2, &p->triangle.n3};
    for (int k = 0; k < 6; k++) std::fwrite(v[k], 4, 3, f);
  }
  for (const Primitive *p: prims) std::fwrite(&p->bsdf->rec, 4, 9, f);
  return std::fclose(f) == 0 ? 0 : 1;
}

int main(int argc, char **argv) {
  const char *bin_out = nullptr;
  if (argc >= 3 && std::string(argv[1]) == "--bin") {
    bin_out = argv[2];
    argv += 2;
    argc -= 2;
  }
  if (argc < 2) return 2;
  Vector3f scale = {1, 1, 1}, translate = {0, 0, 0};
  if (argc >= 8) {
    scale = {(float) atof(argv[2]), (float) atof(argv[3]), (float) atof(argv[4])};
    translate = {(float) atof(argv[5]), (float) atof(argv[6]), (float) atof(argv[7])};
  }
  std::vector<Primitive *> prims;
  if (!Generator::from_obj(&prims, argv[1], nullptr, scale, translate)) return 1;
  if (bin_out) return dump_binary(bin_out, prims);
  for (const Primitive *p: prims) {
    const Vector3f *v[6] = {&p->triangle.p1, &p->triangle.p2, &p->triangle.p3, &p->triangle.n1, &p->triangle.n2, &p->triangle.n3};
    for (int k = 0; k < 6; k++) std::printf("%.9g %.9g %.9g ", v[k]->x, v[k]->y, v[k]->z);
    std::printf("%d", p->bsdf->rec.type);
    for (int k = 0; k < 8; k++) std::printf(" %.9g", p->bsdf->rec.p[k]);
    std::printf("\n");
  }
  return 0;
}
