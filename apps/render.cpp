// Command-line renderer written against the reference's host API (the call sequence of the reference's apps/app.cu:24-30:
// ctor -> set_camera -> resize -> set_scene -> raytrace -> save_to_file), running on libmirt.so.
//   render out.bmp [--obj file.obj] [--w 640 --h 360 --spp 16 --depth 8] [--rng 0|1] [--device-build]
//              [--gpus N | --parts N] [--partition spp|tile]     (N GPUs, or N logical partitions of GPU 0; one RCCL reduce)
// Without --obj it builds the reference's Generator::room plus a few triangles, so it runs with no assets.
#include <cstdlib>
#include <cstring>
#include <string>

#include "raytracer_amd/pathtracer.hpp"

int main(int argc, char **argv) {
  std::string out = "out.bmp", obj;
  uint w = 640, h = 360, spp = 16, depth = 8, rng = 0, device_build = 0, gpus = 1, parts = 1;
  std::string partition = "spp";
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() { return i + 1 < argc ? argv[++i] : (char *) "0"; };
    if (a == "--obj") obj = next();
    else if (a == "--w") w = (uint) atoi(next());
    else if (a == "--h") h = (uint) atoi(next());
    else if (a == "--spp") spp = (uint) atoi(next());
    else if (a == "--depth") depth = (uint) atoi(next());
    else if (a == "--rng") rng = (uint) atoi(next());
    else if (a == "--device-build") device_build = 2;
    else if (a == "--gpus") gpus = (uint) atoi(next());
    else if (a == "--parts") parts = (uint) atoi(next());
    else if (a == "--partition") partition = next();
    else out = a;
  }
  std::vector<Primitive *> prims;
  static BSDF light_bsdf = BSDFFactory::createEmission({4, 4, 4});
  Generator::room(&prims, 4, 3, 4, &white_bsdf, &white_bsdf, &red_bsdf, &green_bsdf, &white_bsdf, &white_bsdf);
  if (!obj.empty()) {
    if (!Generator::from_obj(&prims, obj, nullptr, {1, 1, 1}, {0, 0, 0})) return 1;
  } else {
    prims.push_back(new Primitive(PrimitiveFactory::createTriangle({-.8f, 0, .5f}, {.8f, 0, .5f}, {0, 1.6f, .2f}, {0, .2f, -1},
                                                                   {0, .2f, -1}, {0, .2f, -1}, &mirror_bsdf)));
    prims.push_back(new Primitive(PrimitiveFactory::createTriangle({-.5f, 2.999f, -.5f}, {.5f, 2.999f, -.5f}, {.5f, 2.999f, .5f},
                                                                   {0, -1, 0}, {0, -1, 0}, {0, -1, 0}, &light_bsdf)));
  }
  std::vector<SceneLight> lights = {SceneLightFactory::create_area({12, 12, 12}, {0, 2.99f, 0}, {0, -1, 0}, {1, 0, 0}, {0, 0, 1}),
                                    SceneLightFactory::create_point({1.5f, 1.5f, 1.5f}, {1.5f, 2.5f, -1.5f})};
  PathTracer tracer({spp, 1, spp + 1, 0.05f, depth, 4});
  tracer.options.rng_mode = rng;
  tracer.build_options.own_tree_builder = device_build;
  for (uint k = 0; k < gpus && gpus > 1; k++) tracer.devices.push_back((int) k);
  for (uint k = 0; k < parts && parts > 1; k++) tracer.devices.push_back(0);
  tracer.partition = partition == "tile" ? MIRT_PARTITION_TILE : MIRT_PARTITION_SPP;
  tracer.set_camera({0, 1.5f, -1.95f}, {0, 1.2f, 0}, {0, 1, 0}, 70, 70.f * h / w, 0, INFINITY);
  tracer.resize(w, h);
  tracer.set_scene(prims, lights);
  tracer.raytrace();
  tracer.save_to_file(out);
  std::printf("%llu rays, %.2f ms on the GPU\n", (unsigned long long) (tracer.stats.closest_rays + tracer.stats.shadow_rays),
              tracer.stats.render_ms);
  return 0;
}
