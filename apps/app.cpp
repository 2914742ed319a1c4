// The reference's application (apps/app.cu:8-38) against the MI355X render path: 500 000 emissive particles thrown from the
// origin, 80 frames of physics + scene rebuild + render + BMP, written against the same host API (PhysicsSystem, Generator,
// PathTracer).  The reference takes no arguments; with none this runs its exact configuration.  Optional:
//   app [--frames 80] [--particles 500000] [--w 800 --h 600] [--spp 100] [--prefix sim] [--no-save] [--host-build] [--fast]
// --fast (an extension, not the reference's image): counter-based RNG, 8 paths per pixel in flight and a share pool of one-sample
// shares (PathTracer::options) instead of the reference's one sequential xorshift stream per pixel; the adaptive stop is off then.
// The traversal's own tree is built on the GPU (PLOC): the scene is rebuilt every frame.
#include <chrono>
#include <cstdlib>
#include <string>

#include "raytracer_amd/physics.hpp"

int main(int argc, char **argv) {
  uint frames = 80, particles = 500000, w = 800, h = 600, spp = 100, builder = 2;
  bool save = true, fast = false;
  std::string prefix = "sim";
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() { return i + 1 < argc ? argv[++i] : (char *) "0"; };
    if (a == "--frames") frames = (uint) atoi(next());
    else if (a == "--particles") particles = (uint) atoi(next());
    else if (a == "--w") w = (uint) atoi(next());
    else if (a == "--h") h = (uint) atoi(next());
    else if (a == "--spp") spp = (uint) atoi(next());
    else if (a == "--prefix") prefix = next();
    else if (a == "--no-save") save = false;
    else if (a == "--host-build") builder = 0;
    else if (a == "--fast") fast = true;
  }
  printf("Starting program\n");

  std::vector<SceneLight> lights;
  lights.push_back(SceneLightFactory::create_directional({.2f, .2f, .2f}, {0, -1, 0}));

  std::vector<Primitive *> primitives;
  PhysicsSystem physics(particles, {0, 0, 0}, {0, 1, 0}, 250);

  const auto t0 = std::chrono::steady_clock::now();
  double gpu_ms = 0;
  unsigned long long rays = 0;
  for (uint i = 0; i < frames; i++) {
    physics.step(.075f);
    physics.to_primitives(&primitives);
    Generator::room(&primitives, 10000, 0, 10000, &white_bsdf, nullptr, nullptr, nullptr, nullptr, nullptr);

    PathTracer pathtracer({spp, 1, 64, 0.05f, 8, 4});
    pathtracer.build_options.own_tree_builder = builder;
    if (fast) {
      pathtracer.options.rng_mode = 1;
      pathtracer.options.streams = spp >= 8 ? 8 : 1;
      pathtracer.options.shares = spp;                     // one share per sample (ignored unless it exceeds the streams)
    }
    pathtracer.set_camera({0, 175, -300}, {0, 20, 0}, {0, 1, 0}, 42, (float) (42 * (int) h / (int) w), 0, INFINITY, 0, 0);
    pathtracer.resize(w, h);
    pathtracer.set_scene(primitives, lights);

    pathtracer.raytrace();
    if (save) pathtracer.save_to_file(prefix + std::to_string(i) + ".bmp");
    gpu_ms += pathtracer.stats.render_ms;
    rays += pathtracer.stats.closest_rays + pathtracer.stats.shadow_rays;

    for (Primitive *item: primitives) {
      if (item->type != Primitive::TRIANGLE) delete item->bsdf;
      delete item;                                            // (the reference leaks the primitives themselves)
    }
    primitives.clear();
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("%u frames of %u particles at %ux%u in %.2f s: %.2f frames/s (physics + scene build + render + BMP), %.1f ms/frame on the "
         "GPU, %.0f Mrays/s\n", frames, particles, w, h, secs, frames / secs, gpu_ms / (frames ? frames : 1), rays / (gpu_ms * 1e3 + 1e-9));
  return 0;
}
