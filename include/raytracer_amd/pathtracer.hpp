// C++ mirror of the reference's host API for the render path (reference: include/raytracer/renderer.cuh:59-230,
// scene/{primitives,bsdf,lights}.cuh factories, geometry/generator.cuh) on top of the C ABI in mirt.h.
// Same class names, member names, argument meaning and error behaviour (precondition failures print and return), so code
// written against the reference's `PathTracer` compiles against this header and renders through libmirt.so on an MI355X.
// Header-only; link with -lmirt.  No reference code is reused: the bodies are calls into the C ABI.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../mirt.h"

#ifndef uint
#define uint unsigned int
#endif
// util/misc.cuh:10-19: the constants an application can name (apps/app.cu passes INF_F as the far clip)
#ifndef PI
#define PI 3.14159265358979323f
#endif
#ifndef EPS_F
#define EPS_F 0.00001f
#endif
#ifndef INF_F
#define INF_F INFINITY
#endif

#include "linalg.hpp"      // Vector3f / Vector2f / Vector3i / Matrix3f of the kept API (linalg/*.cuh)

// scene/bsdf.cuh:31-67 -- the stored record IS the C-ABI record
class BSDF {
public:
  enum { DIFFUSE, EMISSION, REFLECTION, REFRACTION, GLASS, MICROFACET, TEXTURE, INVALID };
  mirt_bsdf rec;
  BSDF() { std::memset(&rec, 0, sizeof rec); rec.type = INVALID; }
};

class BSDFFactory {                                          // scene/bsdf.cuh:311-356
  static BSDF make(int type, std::initializer_list<float> p) {
    BSDF b;
    b.rec.type = type;
    int k = 0;
    for (float v: p) b.rec.p[k++] = v;
    return b;
  }

public:
  static BSDF createDiffuse(const Vector3f &r) { return make(BSDF::DIFFUSE, {r.x, r.y, r.z}); }
  static BSDF createEmission(const Vector3f &r) { return make(BSDF::EMISSION, {r.x, r.y, r.z}); }
  static BSDF createReflection(float roughness, const Vector3f &r) { return make(BSDF::REFLECTION, {roughness, r.x, r.y, r.z}); }
  static BSDF createRefraction(float ior, float roughness, const Vector3f &t) {
    return make(BSDF::REFRACTION, {ior, roughness, t.x, t.y, t.z});
  }
  static BSDF createGlass(float ior, float roughness, const Vector3f &r, const Vector3f &t) {
    return make(BSDF::GLASS, {ior, roughness, r.x, r.y, r.z, t.x, t.y, t.z});
  }
  static BSDF createMicrofacet(float alpha, const Vector3f &eta, const Vector3f &k) {
    return make(BSDF::MICROFACET, {alpha, eta.x, eta.y, eta.z, k.x, k.y, k.z});
  }
};

// the 19 presets of scene/bsdf.cuh:358-388, same names.  The conductors are measured optical constants (eta, k per RGB
// channel): data, restated as a table.
static BSDF mirror_bsdf = BSDFFactory::createReflection(0.f, {.9f, .9f, .9f});
static BSDF refract_bsdf = BSDFFactory::createRefraction(1.45f, 0.f, {1, 1, 1});
static BSDF glass_bsdf = BSDFFactory::createGlass(1.45f, 0.f, {1, 1, 1}, {1, 1, 1});
#define MIRT_CONDUCTOR(name, alpha, er, eg, eb, kr, kg, kb) \
  static BSDF name##_bsdf = BSDFFactory::createMicrofacet((float) (alpha), {(float) (er), (float) (eg), (float) (eb)}, {(float) (kr), (float) (kg), (float) (kb)})
MIRT_CONDUCTOR(aluminum, 0.5, 1.34560, 0.96521, 0.61722, 7.47460, 6.39950, 5.30310);
MIRT_CONDUCTOR(brass, 0.5, 0.44400, 0.52700, 1.09400, 3.69500, 2.76500, 1.82900);
MIRT_CONDUCTOR(copper, 0.5, 0.27105, 0.67693, 1.31640, 3.60920, 2.62480, 2.29210);
MIRT_CONDUCTOR(gold, 0.05, 0.18299, 0.42108, 1.37340, 3.42420, 2.34590, 1.77040);
MIRT_CONDUCTOR(iron, 0.5, 2.91140, 2.94970, 2.58450, 3.08930, 2.93180, 2.76700);
MIRT_CONDUCTOR(lead, 0.5, 1.91000, 1.83000, 1.44000, 3.51000, 3.40000, 3.18000);
MIRT_CONDUCTOR(mercury, 0.5, 2.07330, 1.55230, 1.06060, 5.33830, 4.65100, 3.86280);
MIRT_CONDUCTOR(platinum, 0.5, 2.37570, 2.08470, 1.84530, 4.26550, 3.71530, 3.13650);
MIRT_CONDUCTOR(silver, 0.5, 0.15943, 0.14512, 0.13547, 3.92910, 3.19000, 2.38080);
MIRT_CONDUCTOR(titanium, 0.5, 2.74070, 2.54180, 2.26700, 3.81430, 3.43450, 3.03850);
MIRT_CONDUCTOR(blue_metal, 0.5, 2.5, .2, .05, 0.5, 0.5, 0.5);
#undef MIRT_CONDUCTOR
static BSDF black_bsdf = BSDFFactory::createDiffuse({0, 0, 0});
static BSDF white_bsdf = BSDFFactory::createDiffuse({.6f, .6f, .6f});
static BSDF red_bsdf = BSDFFactory::createDiffuse({.6f, .2f, .2f});
static BSDF green_bsdf = BSDFFactory::createDiffuse({.2f, .6f, .2f});
static BSDF blue_bsdf = BSDFFactory::createDiffuse({.2f, .2f, .6f});

struct Primitive {                                           // scene/primitives.cuh:10-27
  enum { INVALID, SPHERE, TRIANGLE } type = INVALID;
  BSDF *bsdf = nullptr;
  struct { Vector3f origin; float r = 0, r2 = 0; } sphere;
  struct { Vector3f p1, p2, p3, n1, n2, n3; } triangle;
};

class PrimitiveFactory {                                     // scene/primitives.cuh:182-201
public:
  static Primitive createSphere(const Vector3f &c, float radius, BSDF *bsdf) {
    Primitive p;
    p.type = Primitive::SPHERE;
    p.bsdf = bsdf;
    p.sphere.origin = c; p.sphere.r = radius; p.sphere.r2 = radius * radius;
    return p;
  }
  static Primitive createTriangle(const Vector3f &p1, const Vector3f &p2, const Vector3f &p3, const Vector3f &n1,
                                  const Vector3f &n2, const Vector3f &n3, BSDF *bsdf) {
    Primitive p;
    p.type = Primitive::TRIANGLE;
    p.bsdf = bsdf;
    p.triangle.p1 = p1; p.triangle.p2 = p2; p.triangle.p3 = p3;
    p.triangle.n1 = n1; p.triangle.n2 = n2; p.triangle.n3 = n3;
    return p;
  }
};

class SceneLight {                                           // scene/lights.cuh:8-35
public:
  enum { AREA, POINT, DIRECTIONAL, INVALID };
  mirt_light rec;
  SceneLight() { std::memset(&rec, 0, sizeof rec); rec.type = INVALID; }
};

class SceneLightFactory {                                    // scene/lights.cuh:123-146
  static void put(float *d, const Vector3f &v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

public:
  static SceneLight create_area(const Vector3f &radiance, const Vector3f &position, const Vector3f &direction,
                                const Vector3f &dim_x, const Vector3f &dim_y) {
    SceneLight l;
    l.rec.type = SceneLight::AREA;
    put(l.rec.p, radiance); put(l.rec.p + 3, position); put(l.rec.p + 6, direction.unit());
    put(l.rec.p + 9, dim_x); put(l.rec.p + 12, dim_y);
    l.rec.p[15] = dim_x.norm() * dim_y.norm();
    return l;
  }
  static SceneLight create_point(const Vector3f &radiance, const Vector3f &position) {
    SceneLight l;
    l.rec.type = SceneLight::POINT;
    put(l.rec.p, radiance); put(l.rec.p + 3, position);
    return l;
  }
  static SceneLight create_directional(const Vector3f &radiance, const Vector3f &direction) {
    SceneLight l;
    l.rec.type = SceneLight::DIRECTIONAL;
    put(l.rec.p, radiance); put(l.rec.p + 3, (-direction).unit());
    return l;
  }
};

struct Parameters {                                          // raytracer.cuh:10-17 (same order: apps init it positionally)
  uint samples_per_pixel = 1024;
  uint samples_per_light = 1;
  uint samples_per_batch = 64;
  float max_tolerance = 0.005f;
  uint max_ray_depth = 32;
  uint bvh_node_size = 4;
};
static Parameters default_parameters;

class SampleBuffer {                                         // util/image.cuh:9-34
public:
  uint w = 0, h = 0;
  Vector3i *data = nullptr;
  std::vector<Vector3i> store;
  void resize(uint nw, uint nh) {
    w = nw; h = nh;
    store.assign((size_t) w * h, Vector3i());
    data = store.data();
  }
};

struct Camera {
  mirt_camera rec;
};

struct Generator {
  // geometry/generator.cuh:10-96: up to six walls, two triangles each, same vertex order
  static void room(std::vector<Primitive *> *prims, float xs, float ys, float zs, BSDF *floor_bsdf, BSDF *ceiling_bsdf,
                   BSDF *left_bsdf, BSDF *right_bsdf, BSDF *front_bsdf, BSDF *back_bsdf) {
    const float x = xs / 2, y = ys, z = zs / 2;
    auto tri = [&](Vector3f a, Vector3f b, Vector3f c, Vector3f n, BSDF *m) {
      prims->push_back(new Primitive(PrimitiveFactory::createTriangle(a, b, c, n, n, n, m)));
    };
    if (floor_bsdf) { tri({-x, 0, -z}, {x, 0, -z}, {x, 0, z}, {0, 1, 0}, floor_bsdf); tri({-x, 0, -z}, {-x, 0, z}, {x, 0, z}, {0, 1, 0}, floor_bsdf); }
    if (ceiling_bsdf) { tri({-x, y, -z}, {x, y, -z}, {x, y, z}, {0, -1, 0}, ceiling_bsdf); tri({-x, y, -z}, {-x, y, z}, {x, y, z}, {0, -1, 0}, ceiling_bsdf); }
    if (left_bsdf) { tri({-x, 0, -z}, {-x, y, -z}, {-x, y, z}, {1, 0, 0}, left_bsdf); tri({-x, 0, -z}, {-x, 0, z}, {-x, y, z}, {1, 0, 0}, left_bsdf); }
    if (right_bsdf) { tri({x, 0, -z}, {x, y, -z}, {x, y, z}, {-1, 0, 0}, right_bsdf); tri({x, 0, -z}, {x, 0, z}, {x, y, z}, {-1, 0, 0}, right_bsdf); }
    if (back_bsdf) { tri({x, 0, z}, {-x, y, z}, {x, y, z}, {0, 0, -1}, back_bsdf); tri({x, 0, z}, {-x, y, z}, {-x, 0, z}, {0, 0, -1}, back_bsdf); }
    if (front_bsdf) { tri({x, 0, -z}, {-x, y, -z}, {x, y, -z}, {0, 0, 1}, front_bsdf); tri({x, 0, -z}, {-x, y, -z}, {-x, 0, -z}, {0, 0, 1}, front_bsdf); }
  }

  // geometry/generator.cuh:98-180 with an own OBJ/MTL reader (the reference uses rapidobj, absent here: triangle order of
  // polygons with more than three vertices is a fan and is NOT pinned against rapidobj).  MTL -> BSDF rule: :126-148.
  // Divergences, documented: missing normals get the geometric normal (the reference reads out of bounds);
  // faces without a material get a default diffuse instead of a null BSDF.  Defaults of a material's missing statements:
  // zero colours, Ns = 1, Ni = 1, illum 0 - what rapidobj's default-constructed Material holds as far as we know (tinyobj
  // semantics); rapidobj is not in /root/reference, so this is UNPINNED: the bundled .mtl files the tests read
  // (tests/test_obj_ingest.py) spell out every statement the MTL -> BSDF rule looks at.
  static bool from_obj(std::vector<Primitive *> *prims, const std::string &file, BSDF *bsdf = nullptr,
                       Vector3f scale = {1, 1, 1}, Vector3f translate = {0, 0, 0}) {
    std::ifstream in(file);
    if (!in) {
      std::printf("could not open %s\n", file.c_str());
      return false;
    }
    struct Mtl { float Kd[3] = {0, 0, 0}, Ks[3] = {0, 0, 0}, Ke[3] = {0, 0, 0}, Tf[3] = {0, 0, 0}; float Ni = 1, Ns = 1; int illum = 0; };
    std::map<std::string, Mtl> mtls;
    std::map<std::string, BSDF *> made;
    std::vector<Vector3f> P, N;
    std::string dir = file.substr(0, file.find_last_of("/\\") + 1), line, cur;
    auto load_mtl = [&](const std::string &name) {
      std::ifstream m(dir + name);
      std::string l, tag, active;
      while (std::getline(m, l)) {
        std::istringstream s(l);
        if (!(s >> tag)) continue;
        if (tag == "newmtl") { s >> active; mtls[active] = Mtl(); }
        else if (active.empty()) continue;
        else if (tag == "Kd") s >> mtls[active].Kd[0] >> mtls[active].Kd[1] >> mtls[active].Kd[2];
        else if (tag == "Ks") s >> mtls[active].Ks[0] >> mtls[active].Ks[1] >> mtls[active].Ks[2];
        else if (tag == "Ke") s >> mtls[active].Ke[0] >> mtls[active].Ke[1] >> mtls[active].Ke[2];
        else if (tag == "Tf") s >> mtls[active].Tf[0] >> mtls[active].Tf[1] >> mtls[active].Tf[2];
        else if (tag == "Ni") s >> mtls[active].Ni;
        else if (tag == "Ns") s >> mtls[active].Ns;
        else if (tag == "illum") s >> mtls[active].illum;
      }
    };
    auto material = [&]() -> BSDF * {
      if (bsdf) return bsdf;
      if (cur.empty() || !mtls.count(cur)) return &white_bsdf;
      auto it = made.find(cur);
      if (it != made.end()) return it->second;
      const Mtl &m = mtls[cur];
      BSDF *b;
      if (m.Ke[0] > 0) b = new BSDF(BSDFFactory::createEmission({m.Ke[0], m.Ke[1], m.Ke[2]}));
      else if (m.illum == 3) b = &mirror_bsdf;
      else if (m.illum == 4) b = new BSDF(BSDFFactory::createGlass(m.Ni, 0, {m.Tf[0], m.Tf[1], m.Tf[2]}, {m.Tf[0], m.Tf[1], m.Tf[2]}));
      else if (m.illum == 11) b = new BSDF(BSDFFactory::createMicrofacet(m.Ns / 100, {m.Kd[0], m.Kd[1], m.Kd[2]}, {m.Ks[0], m.Ks[1], m.Ks[2]}));
      else b = new BSDF(BSDFFactory::createDiffuse({m.Kd[0], m.Kd[1], m.Kd[2]}));
      made[cur] = b;
      return b;
    };
    while (std::getline(in, line)) {
      std::istringstream s(line);
      std::string tag;
      if (!(s >> tag)) continue;
      if (tag == "v") { Vector3f v; s >> v.x >> v.y >> v.z; P.push_back(v); }
      else if (tag == "vn") { Vector3f v; s >> v.x >> v.y >> v.z; N.push_back(v); }
      else if (tag == "mtllib") { std::string n; s >> n; load_mtl(n); }
      else if (tag == "usemtl") s >> cur;
      else if (tag == "f") {
        std::vector<int> vi, ni;
        std::string tok;
        while (s >> tok) {
          int v = 0, t = 0, n = 0;
          if (std::sscanf(tok.c_str(), "%d/%d/%d", &v, &t, &n) == 3) {}
          else if (std::sscanf(tok.c_str(), "%d//%d", &v, &n) == 2) {}
          else if (std::sscanf(tok.c_str(), "%d/%d", &v, &t) == 2) { n = 0; }
          else { std::sscanf(tok.c_str(), "%d", &v); n = 0; }
          vi.push_back(v < 0 ? (int) P.size() + v : v - 1);
          ni.push_back(n == 0 ? -1 : (n < 0 ? (int) N.size() + n : n - 1));
        }
        for (size_t k = 1; k + 1 < vi.size(); k++) {
          int id[3] = {vi[0], vi[k], vi[k + 1]}, nd[3] = {ni[0], ni[k], ni[k + 1]};
          Vector3f v[3], n[3];
          bool ok = true;
          for (int j = 0; j < 3; j++) {
            if (id[j] < 0 || id[j] >= (int) P.size()) { ok = false; break; }
            v[j] = P[id[j]] * scale;                             // generator.cuh:169-170
            v[j] += translate;
          }
          if (!ok) continue;
          Vector3f e1 = v[1] - v[0], e2 = v[2] - v[0];
          Vector3f g = Vector3f(e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x).unit();
          for (int j = 0; j < 3; j++) n[j] = (nd[j] >= 0 && nd[j] < (int) N.size()) ? N[nd[j]] : g;
          prims->push_back(new Primitive(PrimitiveFactory::createTriangle(v[0], v[1], v[2], n[0], n[1], n[2], material())));
        }
      }
    }
    return true;
  }
};

class PathTracer {                                           // renderer.cuh:59-230
public:
  Parameters parameters;
  std::vector<Primitive *> primitives;
  std::vector<SceneLight> lights;
  std::shared_ptr<Camera> camera;
  std::shared_ptr<SampleBuffer> sample_buffer;
  mirt_options options;                                      // extension: zero = reference behaviour
  mirt_build_options build_options{};                        // extension: zero = host builders only (own_tree_builder = 1:
                                                             // the traversal's own tree is built on the GPU, for frame loops)
  mirt_stats stats;
  // extension: more than one entry = the frame is split over these HIP devices (an index may repeat: logical partitions of
  // one GPU) by sample slices (MIRT_PARTITION_SPP) or interleaved tiles, and summed with one RCCL reduce (mirt_group_*)
  std::vector<int> devices;
  int partition = MIRT_PARTITION_SPP;

  explicit PathTracer(Parameters p = default_parameters, int device = 0) : parameters(p), camera(nullptr) {
    sample_buffer = std::make_shared<SampleBuffer>();
    std::memset(&options, 0, sizeof options);
    std::memset(&stats, 0, sizeof stats);
    if (mirt_init(device) != MIRT_OK) std::printf("%s\n", mirt_last_error());
  }
  ~PathTracer() { mirt_scene_destroy(scene_); mirt_group_destroy(group_); }
  PathTracer(const PathTracer &) = delete;
  PathTracer &operator=(const PathTracer &) = delete;

  void resize(uint width, uint height) const { sample_buffer->resize(width, height); }

  void set_scene(const std::vector<Primitive *> &p, const std::vector<SceneLight> &l) {      // :84-93
    primitives = p;
    lights = l;
    mirt_scene_destroy(scene_);                              // the cached device BVH is dropped, as d_bvh_opt is
    scene_ = nullptr;
    group_scene_ = false;
  }

  void set_camera(const Vector3f &from, const Vector3f &at, const Vector3f &up, float hFov, float vFov, float nClip,
                  float fClip, float aperture = 0, float focal_distance = 0) {               // :95-101
    camera = std::make_shared<Camera>();
    const float f[3] = {from.x, from.y, from.z}, a[3] = {at.x, at.y, at.z}, u[3] = {up.x, up.y, up.z};
    mirt_camera_look_at(f, a, u, hFov, vFov, nClip, fClip, aperture, focal_distance, &camera->rec);
  }

  void raytrace() {                                          // :117-189; same guard messages
    if (!camera) { std::printf("CAMERA NOT SET!"); return; }
    if (primitives.empty()) { std::printf("SCENE NOT SET!"); return; }
    if (lights.empty()) { std::printf("LIGHTS NOT SET!"); return; }
    if (sample_buffer->w == 0 || sample_buffer->h == 0) { std::printf("SAMPLE BUFFER EMPTY"); return; }
    std::printf("Starting raytracing...\n");
    const bool grouped = devices.size() > 1;
    if (grouped && (!group_ || group_devices_ != devices)) {
      mirt_group_destroy(group_);
      group_ = nullptr;
      group_scene_ = false;
      if (mirt_group_create(devices.data(), (int) devices.size(), &group_) != MIRT_OK) {
        std::printf("%s\n", mirt_last_error());
        return;
      }
      group_devices_ = devices;
    }
    if ((grouped ? !group_scene_ : !scene_) && !upload(grouped)) return;
    mirt_params prm = {parameters.samples_per_pixel, parameters.samples_per_light, parameters.samples_per_batch,
                       parameters.max_tolerance, parameters.max_ray_depth, parameters.bvh_node_size};
    std::printf("Raytracing Pixels\n");
    static_assert(sizeof(Vector3i) == 12, "film layout");
    const int rc = grouped ? mirt_group_render(group_, &camera->rec, &prm, &options, partition, sample_buffer->w, sample_buffer->h,
                                               (int32_t *) sample_buffer->data, &stats)
                           : mirt_render(scene_, &camera->rec, &prm, &options, sample_buffer->w, sample_buffer->h,
                                         (int32_t *) sample_buffer->data, &stats);
    if (rc != MIRT_OK) {
      std::fprintf(stderr, "Fatal error: %s\n*** FAILED - ABORTING\n", mirt_last_error());   // util/misc.cuh:33-43
      std::exit(1);
    }
    std::printf("Pixels Raytraced. Elapsed time = %.1f ms\n", stats.total_ms);
  }

  void save_to_file(const std::string &filename) {           // :103-113
    if (mirt_save_bmp(filename.c_str(), (const int32_t *) sample_buffer->data, sample_buffer->w, sample_buffer->h) == MIRT_OK)
      std::printf("Image saved to %s\n", filename.c_str());
    else
      std::printf("%s\n", mirt_last_error());
  }

private:
  mirt_scene *scene_ = nullptr;
  mirt_group *group_ = nullptr;
  std::vector<int> group_devices_;
  bool group_scene_ = false;

  bool upload(bool grouped = false) {
    std::printf("Creating BVH\n");
    std::vector<float> pos, nrm;
    std::vector<uint32_t> mat;
    std::vector<mirt_bsdf> mats;
    std::unordered_map<const BSDF *, uint32_t> ids;          // one device record per distinct BSDF (bvh.cuh:123-137)
    std::vector<uint8_t> ptype;
    for (const Primitive *p: primitives) {
      if (p->type == Primitive::SPHERE) {
        const float rec[9] = {p->sphere.origin.x, p->sphere.origin.y, p->sphere.origin.z, p->sphere.r, 0, 0, 0, 0, 0};
        pos.insert(pos.end(), rec, rec + 9);
        nrm.insert(nrm.end(), 9, 0.f);
        ptype.push_back(1);
      } else if (p->type == Primitive::TRIANGLE) {
        const Vector3f *v[6] = {&p->triangle.p1, &p->triangle.p2, &p->triangle.p3, &p->triangle.n1, &p->triangle.n2, &p->triangle.n3};
        for (int k = 0; k < 3; k++) { pos.push_back(v[k]->x); pos.push_back(v[k]->y); pos.push_back(v[k]->z); }
        for (int k = 3; k < 6; k++) { nrm.push_back(v[k]->x); nrm.push_back(v[k]->y); nrm.push_back(v[k]->z); }
        ptype.push_back(2);
      } else {
        std::printf("INVALID primitive in the scene\n");
        return false;
      }
      const BSDF *b = p->bsdf ? p->bsdf : &white_bsdf;
      auto it = ids.find(b);
      if (it == ids.end()) {
        it = ids.emplace(b, (uint32_t) mats.size()).first;
        mats.push_back(b->rec);
      }
      mat.push_back(it->second);
    }
    std::vector<mirt_light> ls;
    for (const SceneLight &l: lights) ls.push_back(l.rec);
    if (grouped) {
      if (mirt_group_set_scene(group_, pos.data(), nrm.data(), mat.data(), ptype.data(), (uint32_t) mat.size(), mats.data(),
                               (uint32_t) mats.size(), ls.data(), (uint32_t) ls.size(), parameters.bvh_node_size,
                               &build_options) != MIRT_OK) {
        std::printf("%s\n", mirt_last_error());
        return false;
      }
      group_scene_ = true;
      std::printf("Created BVH\nOptimized BVH\nMoved BVH to %d GPU contexts\n", mirt_group_size(group_));
      return true;
    }
    if (mirt_scene_create_opts(pos.data(), nrm.data(), mat.data(), ptype.data(), (uint32_t) mat.size(), mats.data(),
                               (uint32_t) mats.size(), ls.data(), (uint32_t) ls.size(), parameters.bvh_node_size,
                               &build_options, &scene_) != MIRT_OK) {
      std::printf("%s\n", mirt_last_error());
      return false;
    }
    uint32_t info[8];
    mirt_scene_info(scene_, info);
    std::printf("Created BVH. Depth: %u\nOptimized BVH\nMoved BVH to the GPU\n", info[1]);
    return true;
  }
};
