// TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
//
// Harness around the *reference's own headers* (anthonyebiner/raytracer, mounted read-only at
// /root/reference).  It is compiled where the reference lies (see oracle/Makefile, target
// _ref/libref.so); no reference source is copied into this repository and nothing in here stands
// in for a header the image lacks: only `raytracer/raytracer.cuh` and what it includes are used
// (the integrator, BVH, primitives, BSDFs, lights, camera, sampler, film), none of which need
// fmt or rapidobj.  `raytracer/renderer.cuh` (needs fmt) is NOT included; its 14-line CPU pixel
// loop (`raytrace_cpu`, renderer.cuh:24-37) is re-driven below by calling the reference's
// `random_float` and `fill_color` with the same seed procedure.
//
// Everything is exposed through a flat C ABI so tests can drive it with ctypes and compare it
// with (a) the C restatement in oracle/oracle.c and (b) the HIP product path.
//
// Build variants (oracle/Makefile):
//   libref.so       g++ semantics as-is (unqualified abs(float) -> int abs)
//   libref_fabs.so  same headers, with `using std::abs;` in this TU before the headers so that
//                   abs(float) keeps float semantics as it does under nvcc.
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#ifdef REF_FLOAT_ABS
using std::abs;
#endif

#include "raytracer/raytracer.cuh"
#include "raytracer/util/bitmap_image.cuh"

namespace {

struct RefScene {
  std::vector<BSDF> bsdf_copies;        // one BSDF object per primitive: (prim.bsdf - base) == original id
  std::vector<Primitive> prim_store;
  std::vector<Primitive *> prim_ptrs;
  std::vector<SceneLight> lights;
  BVHAccelOpt *bvh = nullptr;
  Camera *camera = nullptr;
  uint depth = 0;
};

BSDF make_bsdf(const float *m) {
  // m[0] = type (as float), m[1..8] = the union's floats in declaration order (bsdf.cuh:44-67)
  BSDF b;
  int type = (int) m[0];
  switch (type) {
    case 0: return BSDFFactory::createDiffuse({m[1], m[2], m[3]});
    case 1: return BSDFFactory::createEmission({m[1], m[2], m[3]});
    case 2: return BSDFFactory::createReflection(m[1], {m[2], m[3], m[4]});
    case 3: return BSDFFactory::createRefraction(m[1], m[2], {m[3], m[4], m[5]});
    case 4: return BSDFFactory::createGlass(m[1], m[2], {m[3], m[4], m[5]}, {m[6], m[7], m[8]});
    case 5: return BSDFFactory::createMicrofacet(m[1], {m[2], m[3], m[4]}, {m[5], m[6], m[7]});
    default: return b;
  }
}

SceneLight make_light(const float *l) {
  // l[0] = type; AREA: radiance, position, direction, dim_x, dim_y ; POINT: radiance, position ;
  // DIRECTIONAL: radiance, direction the light travels (factory negates+normalises, lights.cuh:140-145)
  int type = (int) l[0];
  const float *p = l + 1;
  switch (type) {
    case 0:
      return SceneLightFactory::create_area({p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]},
                                            {p[9], p[10], p[11]}, {p[12], p[13], p[14]});
    case 1:
      return SceneLightFactory::create_point({p[0], p[1], p[2]}, {p[3], p[4], p[5]});
    default:
      return SceneLightFactory::create_directional({p[0], p[1], p[2]}, {p[3], p[4], p[5]});
  }
}

Ray make_ray(const float *r) {
  // r = origin(3), ctor direction argument(3), min_t, max_t  -> goes through the reference ctor
  Ray ray(Vector3f(r[0], r[1], r[2]), Vector3f(r[3], r[4], r[5]));
  ray.min_t = r[6];
  ray.max_t = r[7];
  return ray;
}

uint32_t fbits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

template<class F>
void parallel_rows(uint h, int nthreads, F f) {
  if (nthreads <= 1) {
    for (uint y = 0; y < h; y++) f(y);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([=]() { for (uint y = t; y < h; y += nthreads) f(y); });
  for (auto &t: th) t.join();
}

}  // namespace

extern "C" {

const char *ref_variant() {
#ifdef REF_FLOAT_ABS
  return "reference headers, g++, float abs";
#else
  return "reference headers, g++, as-is";
#endif
}

// ptype: NULL = all triangles; else per primitive 1 = sphere (pos[0..2] centre, pos[3] radius), 2 = triangle.
void *ref_scene_create(const float *pos, const float *nrm, const uint32_t *mat, const uint8_t *ptype, uint32_t n,
                       const float *mats /* n_mats x 9 */, uint32_t n_mats,
                       const float *lights /* n_lights x 16 */, uint32_t n_lights, uint32_t leaf_size) {
  auto *s = new RefScene();
  s->bsdf_copies.resize(n);
  s->prim_store.resize(n);
  s->prim_ptrs.resize(n);
  for (uint32_t i = 0; i < n; i++) {
    uint32_t m = mat[i] < n_mats ? mat[i] : 0;
    s->bsdf_copies[i] = make_bsdf(mats + 9 * m);
    const float *p = pos + 9 * i, *q = nrm + 9 * i;
    if (ptype && ptype[i] == 1)
      s->prim_store[i] = PrimitiveFactory::createSphere({p[0], p[1], p[2]}, p[3], &s->bsdf_copies[i]);
    else
        s->prim_store[i] = PrimitiveFactory::createTriangle({p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]},
                                                          {q[0], q[1], q[2]}, {q[3], q[4], q[5]}, {q[6], q[7], q[8]},
                                                          &s->bsdf_copies[i]);
    s->prim_ptrs[i] = &s->prim_store[i];
  }
  for (uint32_t i = 0; i < n_lights; i++) s->lights.push_back(make_light(lights + 16 * i));
  {
    BVHAccel bvh(s->prim_ptrs, leaf_size);               // bvh.cuh:177-225
    s->depth = bvh.get_max_depth();
    s->bvh = new BVHAccelOpt(bvh.to_optimized_bvh());    // bvh.cuh:231-244
  }
  return s;
}

void ref_scene_destroy(void *h) {
  auto *s = (RefScene *) h;
  if (!s) return;
  if (s->bvh) {
    free(s->bvh->primitives);
    free(s->bvh->nodes);
    s->bvh->primitives = nullptr;
    s->bvh->nodes = nullptr;
    delete s->bvh;
  }
  delete s->camera;
  delete s;
}

void ref_scene_info(void *h, uint32_t *n_nodes, uint32_t *depth) {
  auto *s = (RefScene *) h;
  *n_nodes = s->bvh->num_nodes;
  *depth = s->depth;
}

// nodes_out: num_nodes x 8 dwords (minp, maxp, w0, w1) exactly as BVHNodeOpt lies in memory;
// flat_orig: for every flat primitive slot, the index the primitive had in the input arrays.
void ref_scene_bvh(void *h, uint32_t *nodes_out, uint32_t *flat_orig) {
  auto *s = (RefScene *) h;
  static_assert(sizeof(BVHNodeOpt) == 32, "BVHNodeOpt");
  memcpy(nodes_out, s->bvh->nodes, (size_t) s->bvh->num_nodes * 32);
  for (uint i = 0; i < s->bvh->num_primitives; i++)
    flat_orig[i] = (uint32_t) (s->bvh->primitives[i].bsdf - s->bsdf_copies.data());
}

void ref_set_camera(void *h, const float *from, const float *at, const float *up, float hfov, float vfov,
                    float nclip, float fclip, float aperture, float focal) {
  auto *s = (RefScene *) h;
  delete s->camera;
  Vector3f f(from[0], from[1], from[2]), a(at[0], at[1], at[2]), u(up[0], up[1], up[2]);
  if (focal == 0) focal = (f - a).norm();                // renderer.cuh:97-99
  s->camera = new Camera(f, a, u, hfov, vfov, nclip, fclip, aperture, focal);
}

// camera_out: origin(3), c2w columns (9), hFov, vFov, nClip, fClip, aperture, focal_distance  (= Camera, 72 B)
void ref_get_camera(void *h, float *camera_out) {
  auto *s = (RefScene *) h;
  static_assert(sizeof(Camera) == 72, "Camera");
  memcpy(camera_out, s->camera, 72);
}

// The reference CPU pixel loop (renderer.cuh:24-37) over rows [0,h), optionally on several threads
// (pixels are independent: each owns its seed).  rgb_out = w*h*3 ints, row-major, y down.
double ref_render(void *h, const uint32_t *params /* spp spl batch depth leaf */, float tol,
                  uint32_t w, uint32_t hgt, int32_t *rgb_out, int nthreads) {
  auto *s = (RefScene *) h;
  Parameters prm;
  prm.samples_per_pixel = params[0];
  prm.samples_per_light = params[1];
  prm.samples_per_batch = params[2];
  prm.max_tolerance = tol;
  prm.max_ray_depth = params[3];
  prm.bvh_node_size = params[4];
  Scene scene = {s->bvh, s->lights.data(), (uint) s->lights.size(), s->camera};
  SampleBuffer buf(w, hgt);
  auto t0 = std::chrono::steady_clock::now();
  parallel_rows(hgt, nthreads, [&](uint y) {
    Parameters p = prm;
    for (uint x = 0; x < w; x++) {
      uint seed = 0x42ffffff & (y * buf.w + x);
      for (uint _ = 0; _ < 10; _++) random_float(&seed);
      fill_color(x, y, &scene, &p, &buf, &seed);
    }
  });
  auto t1 = std::chrono::steady_clock::now();
  for (size_t i = 0; i < (size_t) w * hgt; i++) {
    rgb_out[3 * i + 0] = buf.data[i].x;
    rgb_out[3 * i + 1] = buf.data[i].y;
    rgb_out[3 * i + 2] = buf.data[i].z;
  }
  free(buf.data);
  return std::chrono::duration<double>(t1 - t0).count();
}

// Sample-0 primary ray of every pixel with the reference seed procedure, traced with
// BVHAccelOpt::intersect.  idx = flat primitive index or -1; tbits = bits of ray.max_t after the call.
void ref_primary_hits(void *h, uint32_t w, uint32_t hgt, int32_t *idx_out, uint32_t *tbits_out, float *ray_out) {
  auto *s = (RefScene *) h;
  for (uint y = 0; y < hgt; y++)
    for (uint x = 0; x < w; x++) {
      uint seed = 0x42ffffff & (y * w + x);
      for (uint _ = 0; _ < 10; _++) random_float(&seed);
      Vector2f origin = {float(x), float(y)};
      auto point = (origin + Sampler2D::sample_grid(&seed));
      Ray ray = s->camera->generate_ray(point.x / w, point.y / hgt);
      Intersection isect;
      bool hit = s->bvh->intersect(ray, &isect);
      size_t i = (size_t) y * w + x;
      idx_out[i] = hit ? (int32_t) (isect.primitive - s->bvh->primitives) : -1;
      tbits_out[i] = fbits(ray.max_t);
      if (ray_out) {
        float *r = ray_out + 11 * i;
        r[0] = ray.origin.x; r[1] = ray.origin.y; r[2] = ray.origin.z;
        r[3] = ray.direction.x; r[4] = ray.direction.y; r[5] = ray.direction.z;
        r[6] = ray.inv_d.x; r[7] = ray.inv_d.y; r[8] = ray.inv_d.z;
        r[9] = ray.min_t; r[10] = ray.max_t;
      }
    }
}

// rays: n x 8 floats (origin, ctor direction argument, min_t, max_t).
// out: idx (flat, -1 = miss), tbits (max_t after), nrm (3, unnormalised interpolated normal; closest-hit only)
void ref_trace(void *h, const float *rays, uint32_t n, int anyhit, int32_t *idx_out, uint32_t *tbits_out,
               float *nrm_out) {
  auto *s = (RefScene *) h;
  for (uint32_t i = 0; i < n; i++) {
    Ray ray = make_ray(rays + 8 * i);
    if (anyhit) {
      bool hit = s->bvh->has_intersection(ray);
      idx_out[i] = hit ? 1 : -1;
      tbits_out[i] = fbits(ray.max_t);
    } else {
      Intersection isect;
      bool hit = s->bvh->intersect(ray, &isect);
      idx_out[i] = hit ? (int32_t) (isect.primitive - s->bvh->primitives) : -1;
      tbits_out[i] = fbits(ray.max_t);
      if (nrm_out) {
        nrm_out[3 * i + 0] = hit ? isect.normal.x : 0;
        nrm_out[3 * i + 1] = hit ? isect.normal.y : 0;
        nrm_out[3 * i + 2] = hit ? isect.normal.z : 0;
      }
    }
  }
}

// Float radiance (mean over spp of estimate_global_lighting) per pixel, before tonemap, using the
// reference seed procedure and ray generation of fill_color but no adaptive stop.  For rRMSE tests.
void ref_radiance(void *h, const uint32_t *params, uint32_t w, uint32_t hgt, uint32_t seed_xor, float *rgb_out,
                  int nthreads) {
  auto *s = (RefScene *) h;
  Parameters prm;
  prm.samples_per_pixel = params[0];
  prm.samples_per_light = params[1];
  prm.samples_per_batch = params[2];
  prm.max_ray_depth = params[3];
  prm.bvh_node_size = params[4];
  Scene scene = {s->bvh, s->lights.data(), (uint) s->lights.size(), s->camera};
  parallel_rows(hgt, nthreads, [&](uint y) {
    Parameters p = prm;
    for (uint x = 0; x < w; x++) {
      uint seed = (0x42ffffff & (y * w + x)) ^ seed_xor;
      for (uint _ = 0; _ < 10; _++) random_float(&seed);
      Vector2f origin = {float(x), float(y)};
      Vector3f color = {0, 0, 0};
      for (uint i = 0; i < p.samples_per_pixel; i++) {
        auto point = (origin + Sampler2D::sample_grid(&seed));
        Ray ray = s->camera->generate_ray(point.x / w, point.y / hgt);
        color += estimate_global_lighting(ray, &scene, &p, &seed);
      }
      color /= p.samples_per_pixel;
      float *o = rgb_out + 3 * ((size_t) y * w + x);
      o[0] = color.x; o[1] = color.y; o[2] = color.z;
    }
  });
}

// ---- unit probes -------------------------------------------------------------------------------------------

float ref_random_float(uint32_t *seed) { return random_float(seed); }

int ref_bbox_intersect(const float *minp, const float *maxp, const float *ray8) {
  Ray ray = make_ray(ray8);
  return BBox::intersect(Vector3f(minp[0], minp[1], minp[2]), Vector3f(maxp[0], maxp[1], maxp[2]), ray) ? 1 : 0;
}

// tri: p1 p2 p3 n1 n2 n3 (18 floats).  out: hit flag, max_t bits after, normal(3)
int ref_tri_intersect(const float *tri, const float *ray8, int anyhit, uint32_t *tbits, float *nrm) {
  Ray ray = make_ray(ray8);
  Primitive p = PrimitiveFactory::createTriangle({tri[0], tri[1], tri[2]}, {tri[3], tri[4], tri[5]},
                                                 {tri[6], tri[7], tri[8]}, {tri[9], tri[10], tri[11]},
                                                 {tri[12], tri[13], tri[14]}, {tri[15], tri[16], tri[17]}, nullptr);
  bool hit;
  Intersection isect;
  isect.normal = {0, 0, 0};
  if (anyhit) hit = p.has_intersection(ray);
  else hit = p.intersect(ray, &isect);
  *tbits = fbits(ray.max_t);
  nrm[0] = isect.normal.x; nrm[1] = isect.normal.y; nrm[2] = isect.normal.z;
  return hit ? 1 : 0;
}

void ref_make_coord_space(const float *n, float *o2w9) {
  Matrix3f m;
  make_coord_space(m, Vector3f(n[0], n[1], n[2]));
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) o2w9[3 * c + r] = m[c][r];
}

// isect_out: normal(3) o2w(9 col-major) w2o(9) hit_point(3) o_out(3) = 27 floats
void ref_isect_compute(const float *ray8, float max_t, const float *normal, float *isect_out) {
  Ray ray = make_ray(ray8);
  ray.max_t = max_t;
  Intersection is;
  is.ray = &ray;
  is.primitive = nullptr;
  is.normal = {normal[0], normal[1], normal[2]};
  is.compute();
  float *o = isect_out;
  *o++ = is.normal.x; *o++ = is.normal.y; *o++ = is.normal.z;
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) *o++ = is.o2w[c][r];
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) *o++ = is.w2o[c][r];
  *o++ = is.hit_point.x; *o++ = is.hit_point.y; *o++ = is.hit_point.z;
  *o++ = is.o_out.x; *o++ = is.o_out.y; *o++ = is.o_out.z;
}

void ref_bsdf_f(const float *mat9, const float *wo, const float *wi, float *out3) {
  BSDF b = make_bsdf(mat9);
  Vector3f f = b.f({wo[0], wo[1], wo[2]}, {wi[0], wi[1], wi[2]});
  out3[0] = f.x; out3[1] = f.y; out3[2] = f.z;
}

// out: wi(3) mask(3) pdf(1); returns is_delta
int ref_bsdf_sample(const float *mat9, const float *wo, uint32_t *seed, float *out7) {
  BSDF b = make_bsdf(mat9);
  Vector3f mask(0, 0, 0);
  float pdf = 0;
  Vector3f wi = b.sample({wo[0], wo[1], wo[2]}, &mask, &pdf, seed);
  out7[0] = wi.x; out7[1] = wi.y; out7[2] = wi.z;
  out7[3] = mask.x; out7[4] = mask.y; out7[5] = mask.z;
  out7[6] = pdf;
  Vector3f e = b.get_emission();
  (void) e;
  return b.is_delta() ? 1 : 0;
}

// out: L(3) wi(3) dist pdf
void ref_light_sample(const float *light16, const float *p, uint32_t *seed, float *out8) {
  SceneLight l = make_light(light16);
  Vector3f wi;
  float dist = 0, pdf = 0;
  Vector3f L = l.sample({p[0], p[1], p[2]}, &wi, &dist, &pdf, seed);
  out8[0] = L.x; out8[1] = L.y; out8[2] = L.z;
  out8[3] = wi.x; out8[4] = wi.y; out8[5] = wi.z;
  out8[6] = dist; out8[7] = pdf;
}

// light record as the factory stores it (type + union floats): 17 dwords
void ref_light_record(const float *light16, uint32_t *out17) {
  SceneLight l = make_light(light16);
  static_assert(sizeof(SceneLight) == 68, "SceneLight");
  memcpy(out17, &l, 68);
}

void ref_update_pixel(const float *c, int32_t *out3) {
  SampleBuffer buf(1, 1);
  buf.update_pixel({c[0], c[1], c[2]}, 0, 0);
  out3[0] = buf.data[0].x; out3[1] = buf.data[0].y; out3[2] = buf.data[0].z;
  free(buf.data);
}

// ray_out: origin direction inv_d min_t max_t (11 floats)
void ref_generate_ray(void *h, float x, float y, uint32_t *seed, float *ray_out) {
  auto *s = (RefScene *) h;
  Ray ray = (s->camera->aperture < EPS_F) ? s->camera->generate_ray(x, y)
                                          : s->camera->generate_ray_for_thin_lens(x, y, seed);
  float *r = ray_out;
  r[0] = ray.origin.x; r[1] = ray.origin.y; r[2] = ray.origin.z;
  r[3] = ray.direction.x; r[4] = ray.direction.y; r[5] = ray.direction.z;
  r[6] = ray.inv_d.x; r[7] = ray.inv_d.y; r[8] = ray.inv_d.z;
  r[9] = ray.min_t; r[10] = ray.max_t;
}

// PathTracer::save_to_file (renderer.cuh:103-113) minus its log line.
void ref_save_bmp(const char *path, const int32_t *rgb, uint32_t w, uint32_t h) {
  bitmap_image image(w, h);
  for (uint x = 0; x < w; x++) {
    for (uint y = 0; y < h; y++) {
      Vector3i color(rgb[3 * (y * w + x)], rgb[3 * (y * w + x) + 1], rgb[3 * (y * w + x) + 2]);
      image.set_pixel(x, y, color.x, color.y, color.z);
    }
  }
  image.save_image(path);
}

void ref_sizes(uint32_t *out) {
  out[0] = sizeof(Vector3f); out[1] = sizeof(Matrix3f); out[2] = sizeof(Ray); out[3] = sizeof(Intersection);
  out[4] = sizeof(Primitive); out[5] = sizeof(BSDF); out[6] = sizeof(BVHNodeOpt); out[7] = sizeof(SceneLight);
  out[8] = sizeof(Camera); out[9] = sizeof(Parameters);
}

}  // extern "C"
