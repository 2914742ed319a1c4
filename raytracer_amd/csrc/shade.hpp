// Device-side scene functions of the render path: shading frame, BSDFs, lights, camera.  Each function restates the
// reference function cited beside it with the same operation order and float/double promotions (the reference lets
// unqualified `pow/exp/tan/atan/log/cos/sin/fabs` resolve to the double overloads; `sinf/cosf/...` stay float).
// Transcendentals come from the device libm, so values can differ from the host libm in the last ulp; everything else
// is arithmetic that rounds identically (see device_math.hpp).
#pragma once
#include "device_math.hpp"

namespace mirt {

enum { BSDF_DIFFUSE = 0, BSDF_EMISSION, BSDF_REFLECTION, BSDF_REFRACTION, BSDF_GLASS, BSDF_MICROFACET };
enum { LIGHT_AREA = 0, LIGHT_POINT, LIGHT_DIRECTIONAL };

struct DBsdf {            // = mirt_bsdf, 36 B
  int32_t type;
  float p[8];
};
struct DLight {           // = mirt_light, 68 B
  int32_t type;
  float p[16];
};
struct DCamera {
  V3 origin;
  M3 c2w;
  float tan_h, tan_v;     // tanf(hFov * PI / 360), tanf(vFov * PI / 360): evaluated once on the host (camera.cuh:39)
  float nClip, fClip, aperture, focal_distance;
};

struct DRay {             // what a trace kernel needs: the ctor's arguments (scene/ray.cuh:19-25)
  V3 o, d_arg;
  float min_t, max_t;
};

// ------------------------------------------------------------------------------------------------ scene/bsdf.cuh:9-28
__device__ __forceinline__ void make_coord_space(M3 &o2w, V3 n) {
  V3 z = n, h = n;
  if (fabsf(h.x) <= fabsf(h.y) && fabsf(h.x) <= fabsf(h.z)) h.x = 1.0f;
  else if (fabsf(h.y) <= fabsf(h.x) && fabsf(h.y) <= fabsf(h.z)) h.y = 1.0f;
  else h.z = 1.0f;
  z = vscale(z, 1.0f / vnorm(z));                 // normalize(): *= (1. / norm)
  V3 y = vcross(h, z);
  y = vscale(y, 1.0f / vnorm(y));
  V3 x = vcross(z, y);
  x = vscale(x, 1.0f / vnorm(x));
  o2w.c0 = x; o2w.c1 = y; o2w.c2 = z;
}

struct DIsect {           // scene/ray.cuh:31-50
  V3 normal;
  M3 o2w, w2o;
  V3 hit_point, o_out;
};
__device__ __forceinline__ void isect_compute(DIsect &is, V3 origin, V3 direction, float max_t) {   // ray.cuh:42-49
  is.normal = vunit(is.normal);
  make_coord_space(is.o2w, is.normal);
  is.w2o = mtranspose(is.o2w);
  is.hit_point = vadd(origin, vscale(direction, max_t));
  is.o_out = vunit(mmulv(is.w2o, vneg(direction)));
}

// ------------------------------------------------------------------------------------------------ util/sampler.cuh:64-71
__device__ __forceinline__ V3 sample_cosine_hemisphere(Rng &r) {
  float Xi1 = rng_next(r);
  float Xi2 = rng_next(r);
  float a = acosf(sqrtf(Xi1));
  float b = 2.0f * MIRT_PI * Xi2;
  float sa = sinf(a);
  return mk(sa * cosf(b), sa * sinf(b), cosf(a));
}

// ------------------------------------------------------------------------------------------------ scene/bsdf.cuh
__device__ __forceinline__ bool bsdf_is_delta(int type) {                                            // :124
  return type == BSDF_REFLECTION || type == BSDF_REFRACTION || type == BSDF_GLASS;
}
__device__ __forceinline__ float acos_theta(V3 w) {                                                  // :257
  return acosf(smin(smax(w.z, -1.0f + 1e-5f), 1.0f - 1e-5f));
}
__device__ __noinline__ float mf_lambda(float wz, float alpha) {                                      // :261
  float theta = acos_theta(mk(0, 0, wz));
  float a = 1.0f / (alpha * tanf(theta));
  return 0.5f * (erff(a) - 1.0f + expf(-a * a) / (a * MIRT_PI));
}
__device__ __noinline__ V3 microfacet_f(const DBsdf &b, V3 o_out, V3 o_in) {                          // :157-164
  if (o_out.z <= 0 || o_in.z <= 0) return mk(0, 0, 0);
  float alpha = b.p[0];
  V3 eta = mk(b.p[1], b.p[2], b.p[3]), k = mk(b.p[4], b.p[5], b.p[6]);
  V3 h = vunit(vadd(o_out, o_in));
  // fresnel :277-285
  float t = acos_theta(o_in);
  V3 a = vadd(vmul(eta, eta), vmul(k, k));
  V3 bb = vscale(vlscale(2, eta), (float) cos((double) t));
  float c = (float) pow(cos((double) t), 2.0);
  V3 rs = vdivv(vadd(vsub(a, bb), vsplat(c)), vadd(vadd(a, bb), vsplat(c)));
  V3 rp = vdivv(vadd(vsub(vscale(a, c), bb), vsplat(1)), vadd(vadd(vscale(a, c), bb), vsplat(1)));
  V3 F = vdiv(vadd(rs, rp), 2);
  // shadow_masking :267-269 (double)
  double G = 1.0 / (1.0 + (double) mf_lambda(o_in.z, alpha) + (double) mf_lambda(o_out.z, alpha));
  // ndf :271-275 (double)
  double th = (double) acos_theta(h);
  double a2 = (double) (alpha * alpha);
  double D = exp(-(pow(tan(th), 2.0) / a2)) / ((double) MIRT_PI * a2 * pow(cos(th), 4.0));
  V3 r = vscale(vscale(F, (float) G), (float) D);
  return vdiv(r, 4 * o_out.z * o_in.z);
}
// MF = false: instantiation for kernels that never see a MICROFACET material (k_logic<false> defers those slots)
template<bool MF>
__device__ __forceinline__ V3 bsdf_f(const DBsdf &b, V3 o_out, V3 o_in) {                             // :152-169
  if (b.type == BSDF_DIFFUSE) return vdiv(mk(b.p[0], b.p[1], b.p[2]), MIRT_PI);
  if (MF && b.type == BSDF_MICROFACET) return microfacet_f(b, o_out, o_in);
  return mk(0, 0, 0);
}
__device__ __forceinline__ bool bsdf_refract(V3 wo, V3 &wi, float ior) {                              // :292-307
  float n = wo.z > 0 ? 1 / ior : ior;
  float n2 = powf(n, 2);
  float z2 = powf(wo.z, 2);
  if (1 - n2 * (1 - z2) < 0) return false;
  V3 out = mk(-n * wo.x, -n * wo.y, sqrtf(1 - n2 * (1 - z2)));
  if (wo.z > 0) out.z = -out.z;
  wi = out;
  return true;
}
__device__ __noinline__ V3 microfacet_sample(const DBsdf &b, V3 o_out, V3 &mask, float &pdf, Rng &seed) {   // :227-243
  float alpha = b.p[0];
  float rx = rng_next(seed);
  float ry = rng_next(seed);
  float a2 = (float) pow((double) alpha, 2.0);
  float t = (float) atan(sqrt(-a2 * log((double) (1 - rx))));
  float p = 2 * MIRT_PI * ry;
  V3 h = mk(sinf(t) * cosf(p), sinf(t) * sinf(p), cosf(t));
  V3 o_in = vadd(vneg(o_out), vlscale(2 * vdot(o_out, h), h));
  float pt = (float) (2 * sin((double) t) / (a2 * pow(cos((double) t), 3.0)) * exp(-pow(tan((double) t), 2.0) / a2));
  float pp = 1 / (2 * MIRT_PI);
  float pw = (float) ((pt * pp) / sin((double) t));
  pdf = pw / (4 * vdot(o_in, h));
  mask = microfacet_f(b, o_out, o_in);
  return o_in;
}
template<bool MF>
__device__ __forceinline__ V3 bsdf_sample(const DBsdf &b, V3 o_out, V3 &mask, float &pdf, Rng &seed,
                                          bool glass_int_abs) {                                      // :171-251
  float cos_o_out = o_out.z;
  V3 o_in = mk(0, 0, 0);
  switch (b.type) {
    case BSDF_DIFFUSE:
      o_in = sample_cosine_hemisphere(seed);
      pdf = o_in.z / MIRT_PI;
      mask = vdiv(mk(b.p[0], b.p[1], b.p[2]), MIRT_PI);
      return o_in;
    case BSDF_EMISSION:
      o_in = sample_cosine_hemisphere(seed);
      pdf = 1 / MIRT_PI;
      mask = mk(0, 0, 0);
      return o_in;
    case BSDF_REFLECTION:
      o_in = mk(-o_out.x, -o_out.y, o_out.z);                                                         // :287-290
      pdf = o_in.z;
      mask = mk(b.p[1], b.p[2], b.p[3]);
      return o_in;
    case BSDF_REFRACTION: {
      float ior = b.p[0];
      if (!bsdf_refract(o_out, o_in, ior)) {
        pdf = 0;
        mask = mk(0, 0, 0);
        return o_in;
      }
      float n = cos_o_out > 0 ? 1 / ior : ior;
      pdf = o_in.z;
      mask = vdiv(mk(b.p[2], b.p[3], b.p[4]), powf(n, 2));
      return o_in;
    }
    case BSDF_GLASS: {
      float ior = b.p[0];
      V3 refl = mk(b.p[2], b.p[3], b.p[4]), trans = mk(b.p[5], b.p[6], b.p[7]);
      if (!bsdf_refract(o_out, o_in, ior)) {
        o_in = mk(-o_out.x, -o_out.y, o_out.z);
        pdf = o_in.z;
        mask = refl;
        return o_in;
      }
      float q = (1 - ior) / (1 + ior);
      float r0 = (float) ((double) q * (double) q);                    // pow(x, 2) in double is the exact square
      double base = glass_int_abs ? (double) (1 - abs((int) o_out.z))  // bsdf.cuh:213 under g++
                                  : (double) (1 - fabsf(o_out.z));     // under nvcc: float abs
      // pow(base, 5.0) by multiplication: three double roundings, invisible after the conversion to float except in
      // ~1e-8 of the cases (the device pow is not the host's correctly rounded one either); no double pow in this kernel
      const double base2 = base * base;
      float r = (float) ((double) r0 + (double) (1 - r0) * (base2 * base2 * base));
      if (rng_next(seed) < r) {
        o_in = mk(-o_out.x, -o_out.y, o_out.z);
        pdf = r * o_in.z;
        mask = vlscale(r, refl);
        return o_in;
      }
      float n = cos_o_out > 0 ? 1 / ior : ior;
      pdf = (1 - r) * o_in.z;
      mask = vdiv(vlscale(1 - r, trans), (float) ((double) n * (double) n));
      return o_in;
    }
    case BSDF_MICROFACET:
      if (MF) return microfacet_sample(b, o_out, mask, pdf, seed);
      pdf = 0;
      mask = mk(0, 0, 0);
      return mk(0, 0, 1);
    default:
      pdf = 0;
      mask = mk(0, 0, 0);
      return mk(0, 0, 1);
  }
}

// ------------------------------------------------------------------------------------------------ scene/lights.cuh:74-104
__device__ __forceinline__ V3 light_sample(const DLight &l, V3 hit_point, V3 &o_in, float &dist_to_light, float &pdf,
                                           Rng &seed) {
  const float *p = l.p;
  if (l.type == LIGHT_AREA) {
    float sx = rng_next(seed) - 0.5f;
    float sy = rng_next(seed) - 0.5f;
    V3 d = vsub(vadd(vadd(mk(p[3], p[4], p[5]), vlscale(sx, mk(p[9], p[10], p[11]))), vlscale(sy, mk(p[12], p[13], p[14]))),
                hit_point);
    float dist = vnorm(d);
    float dist2 = dist * dist;                                          // float(pow(double(dist), 2)): exact square, rounded once
    float cos_theta = vdot(d, mk(p[6], p[7], p[8])) / dist;
    o_in = vdiv(d, dist);
    dist_to_light = dist;
    pdf = (float) ((double) dist2 / ((double) p[15] * fabs((double) cos_theta)));
    return cos_theta < 0 ? mk(p[0], p[1], p[2]) : mk(0, 0, 0);
  } else if (l.type == LIGHT_POINT) {
    V3 d = vsub(mk(p[3], p[4], p[5]), hit_point);
    o_in = vunit(d);
    dist_to_light = vnorm(d);
    pdf = 1;
    return mk(p[0], p[1], p[2]);
  }
  o_in = mk(p[3], p[4], p[5]);
  dist_to_light = MIRT_INF;
  pdf = 1;
  return mk(p[0], p[1], p[2]);
}

// ------------------------------------------------------------------------------------------------ scene/camera.cuh:38-60
__device__ __forceinline__ DRay camera_generate_ray(const DCamera &cam, float x, float y, Rng &seed) {
  V3 cv = mk((0.5f - x) * cam.tan_h * 2, (0.5f - y) * cam.tan_v * 2, -1);
  DRay r;
  if (cam.aperture < MIRT_EPS) {                                        // raytracer.cuh:103
    r.o = cam.origin;
    r.d_arg = vunit(mmulv(cam.c2w, cv));
  } else {
    float rndR = rng_next(seed);
    float rndTheta = (float) ((double) rng_next(seed) * 2.0 * (double) MIRT_PI);
    float sr = sqrtf(rndR);
    V3 p_lens = mk(cam.aperture * sr * cosf(rndTheta), cam.aperture * sr * sinf(rndTheta), 0);
    V3 p_focus = vscale(cv, cam.focal_distance);
    r.o = vadd(mmulv(cam.c2w, p_lens), cam.origin);
    r.d_arg = mmulv(cam.c2w, vunit(vsub(p_focus, p_lens)));
  }
  r.min_t = cam.nClip;
  r.max_t = cam.fClip;
  return r;
}

}  // namespace mirt
