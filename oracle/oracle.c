/* TEST INFRASTRUCTURE ONLY -- the CPU oracle.
 *
 * A plain-C restatement of the reference's render path (anthonyebiner/raytracer), function by
 * function, with the reference's operation order, its float/double promotions and its g++
 * (`std::max` / `std::min`) comparison semantics, so that on the same libm it produces the same
 * bits as the reference's own headers compiled with g++ (oracle/ref_harness.cpp -> _ref/libref.so).
 *
 * Pinned by: tests/test_oracle_vs_ref.py (bitwise, when oracle/_ref exists) and the golden vectors
 * under tests/golden/ that oracle/make_golden.py dumped from _ref/libref.so, including the survey's
 * known-answer checksum 950192.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The
 * product path (raytracer_amd/csrc) never does.
 *
 * Extension that is NOT in the reference: rng_mode 1 = the counter-based generator the HIP path uses
 * (Philox2x32-10 keyed by pixel, sample and seed), so the checker can follow the product's sample
 * set; rng_mode 0 = the reference's xorshift32 stream (sampler.cuh:8-15).
 *
 * Citations are file:line under /root/reference/include/raytracer/.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define PI_F (3.14159265358979323f)      /* util/misc.cuh:16 */
#define EPS_F (0.00001f)                 /* util/misc.cuh:17 */
#define MAX_BVH_STACK 32                 /* util/misc.cuh:19 */

/* util/misc.cuh:28-29 : maxf = std::max, minf = std::min on the g++ path */
static inline float maxf(float a, float b) { return (a < b) ? b : a; }
static inline float minf(float a, float b) { return (b < a) ? b : a; }

/* ---------------------------------------------------------------- linalg/Vector3f.cuh ---------- */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;
typedef struct { v3 c[3]; } m3;          /* column major: entries[j] is column j (Matrix3f.cuh:165-171) */

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }          /* :81 */
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }          /* :86 */
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }          /* :91 */
static inline v3 vdivv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }         /* :96 */
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }                               /* :76 */
static inline v3 vscale(v3 a, float c) { return V(a.x * c, a.y * c, a.z * c); }           /* :101 v * c */
static inline v3 vlscale(float c, v3 a) { return V(c * a.x, c * a.y, c * a.z); }          /* :180 c * v */
static inline v3 vdiv(v3 a, float c) {                                                    /* :106-109 */
  const float rc = (float) (1.0 / (double) c);
  return V(rc * a.x, rc * a.y, rc * a.z);
}
static inline v3 vdivassign(v3 a, float c) {                                              /* :133-135 */
  const float rc = (float) (1. / (double) c);
  return V(a.x * rc, a.y * rc, a.z * rc);
}
static inline float vdot(v3 a, v3 b) { return b.x * a.x + b.y * a.y + b.z * a.z; }        /* :50 */
static inline v3 vcross(v3 a, v3 b) {                                                     /* :55 */
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vnorm(v3 a) { return (float) sqrt((double) (a.x * a.x + a.y * a.y + a.z * a.z)); } /* :147 */
static inline v3 vunit(v3 a) { float rn = (float) (1. / (double) vnorm(a)); return vscale(a, rn); } /* :161 */
static inline v3 vrcp(v3 a) {                                                             /* :140 */
  return V((float) (1.0 / (double) a.x), (float) (1.0 / (double) a.y), (float) (1.0 / (double) a.z));
}
static inline float villum(v3 a) { return 0.2126f * a.x + 0.7152f * a.y + 0.0722f * a.z; } /* :173 */
static inline float vget(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline v3 vsplat(float c) { return V(c, c, c); }                                   /* :41 */

static inline v3 mmulv(const m3 *m, v3 x) {                                               /* Matrix3f.cuh:243 */
  return vadd(vadd(vlscale(x.x, m->c[0]), vlscale(x.y, m->c[1])), vlscale(x.z, m->c[2]));
}
static inline m3 mtranspose(const m3 *a) {                                                /* Matrix3f.cuh:130 */
  m3 b;
  b.c[0] = V(a->c[0].x, a->c[1].x, a->c[2].x);
  b.c[1] = V(a->c[0].y, a->c[1].y, a->c[2].y);
  b.c[2] = V(a->c[0].z, a->c[1].z, a->c[2].z);
  return b;
}

/* ---------------------------------------------------------------- util/sampler.cuh ------------- */
typedef struct {
  int mode;          /* 0 = reference xorshift32, 1 = counter (Philox2x32-10) */
  uint32_t state;    /* xorshift state, or the counter n of the next draw */
  uint32_t key;      /* counter mode: stream key */
  uint32_t sample;   /* counter mode: second counter word */
} rng_t;

static inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
static inline void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key, uint32_t *o0, uint32_t *o1) {
  for (int r = 0; r < 10; r++) {
    uint64_t p = (uint64_t) 0xD256D193U * c0;
    uint32_t hi = (uint32_t) (p >> 32), lo = (uint32_t) p;
    c0 = hi ^ key ^ c1;
    c1 = lo;
    key += 0x9E3779B9U;
  }
  *o0 = c0; *o1 = c1;
}
uint32_t orc_ctr_key(uint32_t pixel, uint32_t seed) { return mix32(pixel ^ mix32(seed + 0x9E3779B9U)); }

static inline float u32_to_unit(uint32_t x) {
  /* sampler.cuh:14 : float(x) / float(UINT_MAX); float(UINT_MAX) rounds to 2^32 */
  return (float) x / 4294967296.0f;
}
static inline float random_float(rng_t *r) {                                             /* sampler.cuh:8-15 */
  if (r->mode == 0) {
    uint32_t x = r->state;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    r->state = x;
    return u32_to_unit(x);
  } else {
    uint32_t n = r->state++, o0, o1;
    philox2x32_10(n >> 1, r->sample, r->key, &o0, &o1);
    return u32_to_unit((n & 1) ? o1 : o0);
  }
}
static inline int coin_flip(rng_t *r, float p) { return random_float(r) < p; }           /* :23 */
static inline v2 sample_grid2(rng_t *r) { v2 s; s.x = random_float(r); s.y = random_float(r); return s; } /* :30 */
static inline v3 sample_cosine_hemisphere(rng_t *r) {                                    /* :64-71 */
  float Xi1 = random_float(r);
  float Xi2 = random_float(r);
  float a = acosf(sqrtf(Xi1));
  float b = 2.0f * PI_F * Xi2;
  return V(sinf(a) * cosf(b), sinf(a) * sinf(b), cosf(a));
}

/* ---------------------------------------------------------------- scene/ray.cuh ---------------- */
typedef struct { v3 origin, direction, inv_d; float min_t, max_t; } ray_t;

static inline ray_t make_ray(v3 origin, v3 dir_arg) {                                    /* ray.cuh:19-21 */
  ray_t r;
  r.origin = origin;
  r.direction = vunit(dir_arg);
  r.inv_d = vrcp(dir_arg);            /* of the ctor ARGUMENT, not of the normalised member */
  r.min_t = EPS_F;
  r.max_t = INFINITY;
  return r;
}

/* ---------------------------------------------------------------- scene/bsdf.cuh --------------- */
enum { BSDF_DIFFUSE = 0, BSDF_EMISSION, BSDF_REFLECTION, BSDF_REFRACTION, BSDF_GLASS, BSDF_MICROFACET,
       BSDF_TEXTURE, BSDF_INVALID };
typedef struct { int type; float p[8]; } bsdf_t;    /* 36 B, the union's floats in declaration order (:44-67) */

static int g_glass_float_abs = 0;   /* 0: g++ int abs (reference as compiled here); 1: nvcc-like float abs (Q3) */
void orc_set_glass_float_abs(int v) { g_glass_float_abs = v; }

static void make_coord_space(m3 *o2w, v3 n) {                                            /* bsdf.cuh:9-28 */
  v3 z = n, h = z;
  if (fabsf(h.x) <= fabsf(h.y) && fabsf(h.x) <= fabsf(h.z)) h.x = 1.0f;
  else if (fabsf(h.y) <= fabsf(h.x) && fabsf(h.y) <= fabsf(h.z)) h.y = 1.0f;
  else h.z = 1.0f;
  z = vdivassign(z, vnorm(z));
  v3 y = vcross(h, z);
  y = vdivassign(y, vnorm(y));
  v3 x = vcross(z, y);
  x = vdivassign(x, vnorm(x));
  o2w->c[0] = x; o2w->c[1] = y; o2w->c[2] = z;
}

static inline int bsdf_is_delta(const bsdf_t *b) {                                       /* :124 */
  return b->type == BSDF_REFLECTION || b->type == BSDF_REFRACTION || b->type == BSDF_GLASS;
}
static inline v3 bsdf_emission(const bsdf_t *b) {                                        /* :141 */
  return b->type == BSDF_EMISSION ? V(b->p[0], b->p[1], b->p[2]) : V(0, 0, 0);
}
static inline float acos_theta(v3 w) {                                                   /* :257 */
  return acosf(minf(maxf(w.z, -1.0f + 1e-5f), 1.0f - 1e-5f));
}
static inline float mf_lambda(v3 w, float alpha) {                                       /* :261 */
  float theta = acos_theta(w);
  float a = 1.0f / (alpha * tanf(theta));
  return 0.5f * (erff(a) - 1.0f + expf(-a * a) / (a * PI_F));
}
static inline double mf_shadow_masking(v3 o_out, v3 wi, float alpha) {                   /* :267 */
  return 1.0 / (1.0 + mf_lambda(wi, alpha) + mf_lambda(o_out, alpha));
}
static inline double mf_ndf(v3 h, float alpha) {                                         /* :271 */
  double t = acos_theta(h);
  double a2 = alpha * alpha;
  return exp(-(pow(tan(t), 2) / a2)) / (PI_F * a2 * pow(cos(t), 4));
}
static inline v3 mf_fresnel(v3 o_in, v3 eta, v3 k) {                                     /* :277 */
  float t = acos_theta(o_in);
  v3 a = vadd(vmul(eta, eta), vmul(k, k));
  v3 b = vscale(vlscale(2, eta), (float) cos((double) t));
  float c = (float) pow(cos((double) t), 2);
  v3 rs = vdivv(vadd(vsub(a, b), vsplat(c)), vadd(vadd(a, b), vsplat(c)));
  v3 rp = vdivv(vadd(vsub(vscale(a, c), b), vsplat(1)), vadd(vadd(vscale(a, c), b), vsplat(1)));
  return vdiv(vadd(rs, rp), 2);
}
static v3 bsdf_f(const bsdf_t *b, v3 o_out, v3 o_in) {                                   /* :152 */
  switch (b->type) {
    case BSDF_DIFFUSE:
      return vdiv(V(b->p[0], b->p[1], b->p[2]), PI_F);
    case BSDF_MICROFACET: {
      if (o_out.z <= 0 || o_in.z <= 0) return V(0, 0, 0);
      float alpha = b->p[0];
      v3 eta = V(b->p[1], b->p[2], b->p[3]), k = V(b->p[4], b->p[5], b->p[6]);
      v3 h = vunit(vadd(o_out, o_in));
      v3 r = vscale(vscale(mf_fresnel(o_in, eta, k), (float) mf_shadow_masking(o_out, o_in, alpha)),
                    (float) mf_ndf(h, alpha));
      return vdiv(r, 4 * o_out.z * o_in.z);
    }
    default:
      return V(0, 0, 0);
  }
}
static inline void bsdf_reflect(v3 wo, v3 *wi) { *wi = V(-wo.x, -wo.y, wo.z); }           /* :287 */
static inline int bsdf_refract(v3 wo, v3 *wi, float ior) {                               /* :292 */
  float n = wo.z > 0 ? 1 / ior : ior;
  float n2 = powf(n, 2);
  float z2 = powf(wo.z, 2);
  if (1 - n2 * (1 - z2) < 0) return 0;
  v3 out = V(-n * wo.x, -n * wo.y, sqrtf(1 - n2 * (1 - z2)));
  if (wo.z > 0) out.z = -out.z;
  *wi = out;
  return 1;
}
static v3 bsdf_sample(const bsdf_t *b, v3 o_out, v3 *mask, float *pdf, rng_t *seed) {    /* :171-251 */
  float cos_o_out = o_out.z;
  v3 o_in = V(0, 0, 0);
  switch (b->type) {
    case BSDF_DIFFUSE:
      o_in = sample_cosine_hemisphere(seed);
      *pdf = o_in.z / PI_F;
      *mask = bsdf_f(b, o_out, o_in);
      return o_in;
    case BSDF_EMISSION:
      o_in = sample_cosine_hemisphere(seed);
      *pdf = 1 / PI_F;
      *mask = bsdf_f(b, o_out, o_in);
      return o_in;
    case BSDF_REFLECTION:
      bsdf_reflect(o_out, &o_in);
      *pdf = o_in.z;
      *mask = V(b->p[1], b->p[2], b->p[3]);
      return o_in;
    case BSDF_REFRACTION: {
      float ior = b->p[0];
      if (!bsdf_refract(o_out, &o_in, ior)) {
        *pdf = 0;
        *mask = V(0, 0, 0);
        return o_in;
      }
      float n = cos_o_out > 0 ? 1 / ior : ior;
      *pdf = o_in.z;
      *mask = vdiv(V(b->p[2], b->p[3], b->p[4]), powf(n, 2));
      return o_in;
    }
    case BSDF_GLASS: {
      float ior = b->p[0];
      v3 refl = V(b->p[2], b->p[3], b->p[4]), trans = V(b->p[5], b->p[6], b->p[7]);
      if (!bsdf_refract(o_out, &o_in, ior)) {
        bsdf_reflect(o_out, &o_in);
        *pdf = o_in.z;
        *mask = refl;
        return o_in;
      } else {
        float r0 = (float) pow((double) ((1 - ior) / (1 + ior)), 2);
        float r;
        if (g_glass_float_abs) r = (float) (r0 + (1 - r0) * pow((double) (1 - fabsf(o_out.z)), 5));
        else r = (float) (r0 + (1 - r0) * pow((double) (1 - abs((int) o_out.z)), 5));     /* :213, Q3 */
        if (coin_flip(seed, r)) {
          bsdf_reflect(o_out, &o_in);
          *pdf = r * o_in.z;
          *mask = vlscale(r, refl);
          return o_in;
        } else {
          float n = cos_o_out > 0 ? 1 / ior : ior;
          *pdf = (1 - r) * o_in.z;
          *mask = vdiv(vlscale(1 - r, trans), (float) pow((double) n, 2));
          return o_in;
        }
      }
    }
    case BSDF_MICROFACET: {
      float alpha = b->p[0];
      v2 r = sample_grid2(seed);
      float a2 = (float) pow((double) alpha, 2);
      float t = (float) atan(sqrt(-a2 * log((double) (1 - r.x))));
      float p = 2 * PI_F * r.y;
      v3 h = V(sinf(t) * cosf(p), sinf(t) * sinf(p), cosf(t));
      o_in = vadd(vneg(o_out), vlscale(2 * vdot(o_out, h), h));
      float pt = (float) (2 * sin((double) t) / (a2 * pow(cos((double) t), 3)) * exp(-pow(tan((double) t), 2) / a2));
      float pp = 1 / (2 * PI_F);
      float pw = (float) ((pt * pp) / sin((double) t));
      *pdf = pw / (4 * vdot(o_in, h));
      *mask = bsdf_f(b, o_out, o_in);
      return o_in;
    }
    default:
      *pdf = 0;
      *mask = V(0, 0, 0);
      return V(0, 0, 1);
  }
}

/* ---------------------------------------------------------------- scene/lights.cuh ------------- */
enum { LIGHT_AREA = 0, LIGHT_POINT, LIGHT_DIRECTIONAL, LIGHT_INVALID };
typedef struct { int type; float p[16]; } light_t;   /* 68 B: type + union floats (:18-35) */

/* Builds the stored record from the factory arguments (lights.cuh:123-146). in16: type, then
 * AREA radiance position direction dim_x dim_y | POINT radiance position | DIRECTIONAL radiance travel-dir */
static light_t light_from_args(const float *in16) {
  light_t l;
  memset(&l, 0, sizeof l);
  l.type = (int) in16[0];
  const float *p = in16 + 1;
  if (l.type == LIGHT_AREA) {
    v3 dir = vunit(V(p[6], p[7], p[8]));
    v3 dx = V(p[9], p[10], p[11]), dy = V(p[12], p[13], p[14]);
    memcpy(l.p, p, 6 * sizeof(float));
    l.p[6] = dir.x; l.p[7] = dir.y; l.p[8] = dir.z;
    memcpy(l.p + 9, p + 9, 6 * sizeof(float));
    l.p[15] = vnorm(dx) * vnorm(dy);
  } else if (l.type == LIGHT_POINT) {
    memcpy(l.p, p, 6 * sizeof(float));
  } else {
    l.type = LIGHT_DIRECTIONAL;
    v3 d = vunit(vneg(V(p[3], p[4], p[5])));
    memcpy(l.p, p, 3 * sizeof(float));
    l.p[3] = d.x; l.p[4] = d.y; l.p[5] = d.z;
  }
  return l;
}

static v3 light_sample(const light_t *l, v3 hit_point, v3 *o_in, float *dist_to_light, float *pdf, rng_t *seed) {
  const float *p = l->p;                                                                 /* :74-104 */
  switch (l->type) {
    case LIGHT_AREA: {
      v3 radiance = V(p[0], p[1], p[2]), position = V(p[3], p[4], p[5]), direction = V(p[6], p[7], p[8]);
      v3 dim_x = V(p[9], p[10], p[11]), dim_y = V(p[12], p[13], p[14]);
      float area = p[15];
      v2 s = sample_grid2(seed);
      s.x -= 0.5f; s.y -= 0.5f;
      v3 d = vsub(vadd(vadd(position, vlscale(s.x, dim_x)), vlscale(s.y, dim_y)), hit_point);
      float dist = vnorm(d);
      float dist2 = (float) pow((double) dist, 2);
      float cos_theta = vdot(d, direction) / dist;
      *o_in = vdiv(d, dist);
      *dist_to_light = dist;
      *pdf = (float) ((double) dist2 / ((double) area * fabs((double) cos_theta)));
      return cos_theta < 0 ? radiance : V(0, 0, 0);
    }
    case LIGHT_POINT: {
      v3 d = vsub(V(p[3], p[4], p[5]), hit_point);
      *o_in = vunit(d);
      *dist_to_light = vnorm(d);
      *pdf = 1;
      return V(p[0], p[1], p[2]);
    }
    default:
      *o_in = V(p[3], p[4], p[5]);
      *dist_to_light = INFINITY;
      *pdf = 1;
      return V(p[0], p[1], p[2]);
  }
}

/* ---------------------------------------------------------------- scene/primitives.cuh --------- */
enum { PRIM_INVALID = 0, PRIM_SPHERE = 1, PRIM_TRIANGLE = 2 };
typedef struct {
  int type;
  uint32_t mat;      /* stands for the BSDF pointer */
  uint32_t orig;     /* index in the caller's arrays */
  float d[18];       /* triangle: p1 p2 p3 n1 n2 n3 ; sphere: origin r r2 */
} prim_t;

typedef struct { int hit; uint32_t prim; v3 normal; } isect_raw_t;

static inline v3 P3(const float *f) { return V(f[0], f[1], f[2]); }

static int prim_has_intersection(const prim_t *pr, ray_t *ray) {                          /* :62-107 */
  if (pr->type == PRIM_SPHERE) {
    v3 a = vsub(P3(pr->d), ray->origin);
    float r2 = pr->d[4];
    float b = vdot(a, ray->direction);
    if (b < 0) return 0;
    float c = vdot(a, a) - b * b;
    if (c > r2) return 0;
    float disc = sqrtf(r2 - c);
    float t0 = b - disc, t1 = b + disc;
    if (t1 <= ray->min_t || t0 > ray->max_t || (t0 <= ray->min_t && t1 > ray->max_t)) return 0;
    if (t0 > ray->min_t) ray->max_t = t0;
    else {
      if (t1 > ray->max_t) return 0;
      ray->max_t = t1;
    }
    return 1;
  } else if (pr->type == PRIM_TRIANGLE) {
    v3 p1 = P3(pr->d), p2 = P3(pr->d + 3), p3 = P3(pr->d + 6);
    v3 e1 = vsub(p2, p1), e2 = vsub(p3, p1), s = vsub(ray->origin, p1);
    v3 s1 = vcross(ray->direction, e2), s2 = vcross(s, e1);
    v3 c = vdiv(V(vdot(s2, e2), vdot(s1, s), vdot(s2, ray->direction)), vdot(s1, e1));
    float t = c.x;
    v3 b = V(1 - c.y - c.z, c.y, c.z);
    if (t < 0 || t <= ray->min_t || t > ray->max_t || b.x < 0 || b.y < 0 || b.z < 0) return 0;
    ray->max_t = t;
    return 1;
  }
  return 0;
}

static int prim_intersect(const prim_t *pr, uint32_t idx, ray_t *ray, isect_raw_t *is) {  /* :109-162 */
  if (pr->type == PRIM_SPHERE) {
    v3 a = vsub(P3(pr->d), ray->origin);
    float r2 = pr->d[4];
    float b = vdot(a, ray->direction);
    if (b < 0) return 0;
    float c = vdot(a, a) - b * b;
    if (c > r2) return 0;
    float disc = sqrtf(r2 - c);
    float t0 = b - disc, t1 = b + disc;
    if (t1 <= ray->min_t || t0 > ray->max_t) return 0;
    if (t0 > ray->min_t) ray->max_t = t0;
    else {
      if (t1 > ray->max_t) return 0;
      ray->max_t = t1;
    }
    is->hit = 1; is->prim = idx;
    is->normal = vsub(vadd(ray->origin, vlscale(ray->max_t, ray->direction)), P3(pr->d));
    return 1;
  } else if (pr->type == PRIM_TRIANGLE) {
    v3 p1 = P3(pr->d), p2 = P3(pr->d + 3), p3 = P3(pr->d + 6);
    v3 e1 = vsub(p2, p1), e2 = vsub(p3, p1), s = vsub(ray->origin, p1);
    v3 s1 = vcross(ray->direction, e2), s2 = vcross(s, e1);
    v3 c = vdiv(V(vdot(s2, e2), vdot(s1, s), vdot(s2, ray->direction)), vdot(s1, e1));
    float t = c.x;
    v3 b = V(1 - c.y - c.z, c.y, c.z);
    if (t < 0 || t <= ray->min_t || t > ray->max_t || b.x < 0 || b.y < 0 || b.z < 0) return 0;
    ray->max_t = t;
    is->hit = 1; is->prim = idx;
    is->normal = vadd(vadd(vlscale(b.x, P3(pr->d + 9)), vlscale(b.y, P3(pr->d + 12))), vlscale(b.z, P3(pr->d + 15)));
    return 1;
  }
  return 0;
}

/* ---------------------------------------------------------------- scene/bbox.cuh --------------- */
typedef struct { v3 maxp, minp, extent; } bbox_t;

static inline bbox_t bbox_empty(void) {                                                   /* :13 */
  bbox_t b;
  b.maxp = V(-INFINITY, -INFINITY, -INFINITY);
  b.minp = V(INFINITY, INFINITY, INFINITY);
  b.extent = vsub(b.maxp, b.minp);
  return b;
}
static inline bbox_t bbox_point(v3 p) { bbox_t b; b.minp = p; b.maxp = p; b.extent = vsub(b.maxp, b.minp); return b; }
static inline void bbox_expand_p(bbox_t *b, v3 p) {                                       /* :38 */
  b->minp.x = minf(b->minp.x, p.x); b->minp.y = minf(b->minp.y, p.y); b->minp.z = minf(b->minp.z, p.z);
  b->maxp.x = maxf(b->maxp.x, p.x); b->maxp.y = maxf(b->maxp.y, p.y); b->maxp.z = maxf(b->maxp.z, p.z);
  b->extent = vsub(b->maxp, b->minp);
}
static inline void bbox_expand_b(bbox_t *b, const bbox_t *o) {                            /* :28 */
  b->minp.x = minf(b->minp.x, o->minp.x); b->minp.y = minf(b->minp.y, o->minp.y); b->minp.z = minf(b->minp.z, o->minp.z);
  b->maxp.x = maxf(b->maxp.x, o->maxp.x); b->maxp.y = maxf(b->maxp.y, o->maxp.y); b->maxp.z = maxf(b->maxp.z, o->maxp.z);
  b->extent = vsub(b->maxp, b->minp);
}
static inline v3 bbox_centroid(const bbox_t *b) { return vdiv(vadd(b->minp, b->maxp), 2.0f); } /* :48 */

static inline bbox_t prim_bbox(const prim_t *pr) {                                        /* primitives.cuh:164 */
  if (pr->type == PRIM_SPHERE) {
    bbox_t b;
    float r = pr->d[3];
    b.minp = vsub(P3(pr->d), V(r, r, r));
    b.maxp = vadd(P3(pr->d), V(r, r, r));
    b.extent = vsub(b.maxp, b.minp);
    return b;
  } else if (pr->type == PRIM_TRIANGLE) {
    bbox_t b = bbox_point(P3(pr->d));
    bbox_expand_p(&b, P3(pr->d + 3));
    bbox_expand_p(&b, P3(pr->d + 6));
    return b;
  }
  return bbox_empty();
}

static inline int bbox_intersect(v3 minp, v3 maxp, const ray_t *ray) {                    /* bbox.cuh:67-79 */
  float tmin = 0.0f, tmax = ray->max_t;
  for (int d = 0; d < 3; ++d) {
    float t1 = (vget(minp, d) - vget(ray->origin, d)) * vget(ray->inv_d, d);
    float t2 = (vget(maxp, d) - vget(ray->origin, d)) * vget(ray->inv_d, d);
    tmin = minf(maxf(t1, tmin), maxf(t2, tmin));
    tmax = maxf(minf(t1, tmax), minf(t2, tmax));
  }
  return tmin <= tmax && tmax > ray->min_t;
}

/* ---------------------------------------------------------------- bvh/bvh.cuh ------------------ */
typedef struct { v3 minp, maxp; uint32_t w0, w1; } node_t;   /* 32 B: inner {idx_left, idx_right} | leaf {count|1<<31, idx_start} */

typedef struct bnode { bbox_t bb; struct bnode *l, *r; uint32_t start, end; } bnode_t;

typedef struct {
  prim_t *prims;        /* flat order */
  node_t *nodes;
  uint32_t num_prims, num_nodes, depth;
  bsdf_t *mats; uint32_t num_mats;
  light_t *lights; uint32_t num_lights;
  /* camera (camera.cuh:10-17) */
  v3 cam_origin; m3 c2w; float hFov, vFov, nClip, fClip, aperture, focal_distance;
  int rng_mode; uint32_t seed;
} scene_t;

static bnode_t *construct_bvh(prim_t **v, uint32_t start, uint32_t end, uint32_t max_leaf) { /* :187-225 */
  bbox_t bbox = bbox_empty();
  for (uint32_t p = start; p < end; p++) { bbox_t bb = prim_bbox(v[p]); bbox_expand_b(&bbox, &bb); }
  bnode_t *node = (bnode_t *) calloc(1, sizeof(bnode_t));
  node->bb = bbox;
  if ((int64_t) end - (int64_t) start <= (int64_t) max_leaf) { node->start = start; node->end = end; return node; }
  float ext[3] = {bbox.extent.x, bbox.extent.y, bbox.extent.z};
  int axis = 0;                                   /* std::max_element: first maximum, operator< */
  for (int i = 1; i < 3; i++) if (ext[axis] < ext[i]) axis = i;
  v3 center = V(0, 0, 0);
  for (uint32_t p = start; p < end; p++) { bbox_t bb = prim_bbox(v[p]); center = vadd(center, bbox_centroid(&bb)); }
  center = vdivassign(center, (float) (end - start));
  float cax = vget(center, axis);
  /* libstdc++ std::partition, bidirectional version (stl_algo.h:1487-1510) */
  uint32_t first = start, last = end;
  for (;;) {
    for (;;) {
      if (first == last) goto done;
      bbox_t bb = prim_bbox(v[first]);
      if (vget(bbox_centroid(&bb), axis) < cax) ++first; else break;
    }
    --last;
    for (;;) {
      if (first == last) goto done;
      bbox_t bb = prim_bbox(v[last]);
      if (!(vget(bbox_centroid(&bb), axis) < cax)) --last; else break;
    }
    { prim_t *t = v[first]; v[first] = v[last]; v[last] = t; }
    ++first;
  }
done:;
  uint32_t middle = first;
  if (middle - start == 0 || end - middle == 0) { node->start = start; node->end = end; return node; }
  node->l = construct_bvh(v, start, middle, max_leaf);
  node->r = construct_bvh(v, middle, end, max_leaf);
  return node;
}
static uint32_t count_nodes(const bnode_t *n) { return (!n->l && !n->r) ? 1 : 1 + count_nodes(n->l) + count_nodes(n->r); }
static uint32_t max_depth(const bnode_t *n, uint32_t d) {
  if (!n->l && !n->r) return d;
  uint32_t a = max_depth(n->l, d + 1), b = max_depth(n->r, d + 1);
  return a < b ? b : a;
}
static void free_tree(bnode_t *n) { if (!n) return; free_tree(n->l); free_tree(n->r); free(n); }

static void flatten(const bnode_t *node, prim_t **v, prim_t *prims, node_t *nodes, uint32_t *prim_start,
                    uint32_t *node_start) {                                               /* :272-290 */
  uint32_t n_idx = *node_start;
  nodes[n_idx].minp = node->bb.minp;
  nodes[n_idx].maxp = node->bb.maxp;
  if (!node->l && !node->r) {
    nodes[n_idx].w0 = 0x80000000U | (node->end - node->start);
    nodes[n_idx].w1 = *prim_start;
    for (uint32_t p = node->start; p < node->end; p++) { prims[*prim_start] = *v[p]; (*prim_start)++; }
  } else {
    nodes[n_idx].w1 = ++(*node_start);                      /* idx_right */
    flatten(node->r, v, prims, nodes, prim_start, node_start);
    nodes[n_idx].w0 = ++(*node_start);                      /* idx_left */
    flatten(node->l, v, prims, nodes, prim_start, node_start);
  }
}

static int bvh_has_intersection(const scene_t *s, ray_t *ray, uint64_t *nbox, uint64_t *ntri) { /* :49-76 */
  uint32_t stack[MAX_BVH_STACK], sp = 0;
  stack[sp++] = 0;
  while (sp) {
    node_t node = s->nodes[stack[--sp]];
    if (nbox) (*nbox)++;
    if (!bbox_intersect(node.minp, node.maxp, ray)) continue;
    if (node.w0 & 0x80000000U) {
      for (uint32_t i = node.w1; i < node.w1 + (node.w0 & 0x7fffffffU); i++) {
        if (ntri) (*ntri)++;
        if (prim_has_intersection(&s->prims[i], ray)) return 1;
      }
    } else {
      stack[sp++] = node.w0;
      stack[sp++] = node.w1;
    }
  }
  return 0;
}
static int bvh_intersect(const scene_t *s, ray_t *ray, isect_raw_t *is, uint64_t *nbox, uint64_t *ntri) { /* :78-107 */
  uint32_t stack[MAX_BVH_STACK], sp = 0;
  stack[sp++] = 0;
  int hit = 0;
  while (sp) {
    node_t node = s->nodes[stack[--sp]];
    if (nbox) (*nbox)++;
    if (!bbox_intersect(node.minp, node.maxp, ray)) continue;
    if (node.w0 & 0x80000000U) {
      for (uint32_t i = node.w1; i < node.w1 + (node.w0 & 0x7fffffffU); i++) {
        if (ntri) (*ntri)++;
        int h1 = prim_intersect(&s->prims[i], i, ray, is);
        hit = h1 || hit;
      }
    } else {
      stack[sp++] = node.w0;
      stack[sp++] = node.w1;
    }
  }
  return hit;
}

/* ---------------------------------------------------------------- scene/camera.cuh ------------- */
static ray_t camera_generate_ray(const scene_t *s, float x, float y) {                    /* :38-44 */
  v3 cv = V((0.5f - x) * tanf(s->hFov * PI_F / 360) * 2, (0.5f - y) * tanf(s->vFov * PI_F / 360) * 2, -1);
  ray_t r = make_ray(s->cam_origin, vunit(mmulv(&s->c2w, cv)));
  r.min_t = s->nClip;
  r.max_t = s->fClip;
  return r;
}
static ray_t camera_generate_ray_thin_lens(const scene_t *s, float x, float y, rng_t *seed) { /* :46-60 */
  v2 rnd = sample_grid2(seed);
  float rndR = rnd.x;
  float rndTheta = (float) (rnd.y * 2.0 * PI_F);
  v3 cv = V((0.5f - x) * tanf(s->hFov * PI_F / 360) * 2, (0.5f - y) * tanf(s->vFov * PI_F / 360) * 2, -1);
  v3 p_lens = V(s->aperture * sqrtf(rndR) * cosf(rndTheta), s->aperture * sqrtf(rndR) * sinf(rndTheta), 0);
  v3 p_focus = vscale(cv, s->focal_distance);
  ray_t r = make_ray(vadd(mmulv(&s->c2w, p_lens), s->cam_origin), mmulv(&s->c2w, vunit(vsub(p_focus, p_lens))));
  r.min_t = s->nClip;
  r.max_t = s->fClip;
  return r;
}

/* ---------------------------------------------------------------- scene/ray.cuh:31-50 ---------- */
typedef struct { uint32_t prim; v3 normal; m3 o2w, w2o; v3 hit_point, o_out; } isect_t;

static void isect_compute(isect_t *is, const ray_t *ray) {                                /* ray.cuh:42-49 */
  is->normal = vunit(is->normal);
  make_coord_space(&is->o2w, is->normal);
  is->w2o = mtranspose(&is->o2w);
  is->hit_point = vadd(ray->origin, vscale(ray->direction, ray->max_t));
  is->o_out = vunit(mmulv(&is->w2o, vneg(ray->direction)));
}

/* ---------------------------------------------------------------- raytracer.cuh ---------------- */
typedef struct { uint32_t spp, spl, batch; float tol; uint32_t depth, leaf; } params_t;   /* :10-17 */
typedef struct { uint64_t closest, shadow, box, tri; } counters_t;

static v3 estimate_direct_lighting(const scene_t *s, const params_t *prm, const isect_t *is, rng_t *seed,
                                   counters_t *cnt) {                                     /* :30-51 */
  const bsdf_t *bsdf = &s->mats[s->prims[is->prim].mat];
  if (bsdf_is_delta(bsdf)) return V(0, 0, 0);
  v3 to_light;
  float dist, pdf;
  v3 bump = vscale(is->normal, 0.001f);
  v3 color = V(0, 0, 0);
  for (uint32_t i = 0; i < s->num_lights; i++) {
    for (uint32_t k = 0; k < prm->spl; k++) {
      v3 L = light_sample(&s->lights[i], vadd(is->hit_point, bump), &to_light, &dist, &pdf, seed);
      ray_t sh = make_ray(vadd(is->hit_point, bump), to_light);
      sh.max_t = dist - EPS_F;
      cnt->shadow++;
      if (!bvh_has_intersection(s, &sh, &cnt->box, &cnt->tri)) {
        v3 f = bsdf_f(bsdf, is->o_out, mmulv(&is->w2o, to_light));
        float c = maxf(vdot(is->normal, vunit(sh.direction)), 0.f);
        color = vadd(color, vdiv(vdiv(vscale(vmul(f, L), c), pdf), (float) prm->spl));
      }
    }
  }
  return color;
}

static v3 estimate_global_lighting(ray_t *ray, const scene_t *s, const params_t *prm, rng_t *seed,
                                   counters_t *cnt) {                                     /* :54-90 */
  v3 color_mask = V(1, 1, 1), total = V(0, 0, 0);
  for (uint32_t i = 0; i <= prm->depth; i++) {
    isect_raw_t raw;
    raw.hit = 0; raw.prim = 0; raw.normal = V(0, 0, 0);
    cnt->closest++;
    if (!bvh_intersect(s, ray, &raw, &cnt->box, &cnt->tri)) break;
    isect_t is;
    is.prim = raw.prim;
    is.normal = raw.normal;
    isect_compute(&is, ray);
    const bsdf_t *bsdf = &s->mats[s->prims[is.prim].mat];
    total = vadd(total, vmul(color_mask, bsdf_emission(bsdf)));
    if (!bsdf_is_delta(bsdf)) total = vadd(total, vmul(color_mask, estimate_direct_lighting(s, prm, &is, seed, cnt)));
    float cpdf = maxf(color_mask.x, maxf(color_mask.y, color_mask.z));
    if (!coin_flip(seed, cpdf)) break;
    float pdf;
    v3 mask;
    v3 o_in = bsdf_sample(bsdf, is.o_out, &mask, &pdf, seed);
    if (pdf == 0) break;
    v3 bump = vscale(is.normal, 0.001f);
    bump = o_in.z > 0 ? bump : vneg(bump);
    *ray = make_ray(vadd(is.hit_point, bump), mmulv(&is.o2w, o_in));
    color_mask = vdiv(vdiv(vscale(vmul(color_mask, mask), o_in.z), pdf), minf(cpdf, 1.f));
  }
  return total;
}

/* util/image.cuh:36-48 */
static void update_pixel(v3 c, int32_t *out) {
  float gamma = 2.2f, level = 1.0f;
  float one_over_gamma = 1.0f / gamma;
  float exposure = (float) sqrt(pow(2, (double) level));
  out[0] = (int32_t) (255 * maxf(0.f, minf(1.f, powf(c.x * exposure, one_over_gamma))));
  out[1] = (int32_t) (255 * maxf(0.f, minf(1.f, powf(c.y * exposure, one_over_gamma))));
  out[2] = (int32_t) (255 * maxf(0.f, minf(1.f, powf(c.z * exposure, one_over_gamma))));
}

static rng_t pixel_rng(const scene_t *s, uint32_t pixel_index) {
  rng_t r;
  r.mode = s->rng_mode;
  r.key = 0; r.sample = 0;
  if (s->rng_mode == 0) {
    r.state = (0x42ffffffU & pixel_index) ^ s->seed;          /* renderer.cuh:28 (seed = 0: the reference) */
    for (int k = 0; k < 10; k++) random_float(&r);            /* renderer.cuh:29 */
  } else {
    r.state = 0;
    r.key = orc_ctr_key(pixel_index, s->seed);
  }
  return r;
}

/* raytracer.cuh:93-126.  accum (optional) receives sum r,g,b and the divisor; sample_begin/_count select a
 * slice of the sample sequence in counter mode (extension for spp-sliced multi-GPU runs). */
static void fill_color(uint32_t x, uint32_t y, uint32_t w, uint32_t h, const scene_t *s, const params_t *prm,
                       rng_t *seed, int32_t *rgb_out, float *accum_out, uint32_t sample_begin, counters_t *cnt) {
  float ox = (float) x, oy = (float) y;
  v3 color = V(0, 0, 0);
  int i = 1;
  float s1 = 0, s2 = 0;
  while ((uint32_t) i <= prm->spp) {
    if (seed->mode == 1) { seed->state = 0; seed->sample = sample_begin + (uint32_t) (i - 1); }
    v2 g = sample_grid2(seed);
    float px = ox + g.x, py = oy + g.y;
    ray_t ray;
    if (s->aperture < EPS_F) ray = camera_generate_ray(s, px / (float) w, py / (float) h);
    else ray = camera_generate_ray_thin_lens(s, px / (float) w, py / (float) h, seed);
    v3 sc = estimate_global_lighting(&ray, s, prm, seed, cnt);
    color = vadd(color, sc);
    float illum = villum(sc);
    s1 += illum;
    s2 += powf(illum, 2);
    if ((uint32_t) i % prm->batch == 0) {
      float variance = (float) (((double) s1 - pow((double) s1, 2) / i) / (i - 1));
      float mean = s1 / i;
      if (1.96 * sqrtf(variance / i) <= (double) (prm->tol * mean)) break;
    }
    i++;
  }
  (void) s2;
  if (accum_out) { accum_out[0] = color.x; accum_out[1] = color.y; accum_out[2] = color.z; accum_out[3] = (float) (i - 1); }
  color = vdivassign(color, (float) (i - 1));
  if (rgb_out) update_pixel(color, rgb_out);
}

/* ================================================================ C ABI for the tests ========== */

void *orc_scene_create(const float *pos, const float *nrm, const uint32_t *mat, const uint8_t *ptype, uint32_t n,
                       const float *mats9, uint32_t n_mats, const float *lights16, uint32_t n_lights,
                       uint32_t leaf_size) {
  scene_t *s = (scene_t *) calloc(1, sizeof(scene_t));
  prim_t *store = (prim_t *) calloc(n ? n : 1, sizeof(prim_t));
  prim_t **v = (prim_t **) calloc(n ? n : 1, sizeof(prim_t *));
  for (uint32_t i = 0; i < n; i++) {
    prim_t *p = &store[i];
    p->mat = mat[i] < n_mats ? mat[i] : 0;
    p->orig = i;
    if (ptype && ptype[i] == 1) {
      p->type = PRIM_SPHERE;
      memcpy(p->d, pos + 9 * i, 4 * sizeof(float));
      p->d[4] = p->d[3] * p->d[3];                                  /* primitives.cuh:188 */
    } else {
      p->type = PRIM_TRIANGLE;
      memcpy(p->d, pos + 9 * i, 9 * sizeof(float));
      memcpy(p->d + 9, nrm + 9 * i, 9 * sizeof(float));
    }
    v[i] = p;
  }
  s->mats = (bsdf_t *) calloc(n_mats ? n_mats : 1, sizeof(bsdf_t));
  for (uint32_t i = 0; i < n_mats; i++) {
    s->mats[i].type = (int) mats9[9 * i];
    memcpy(s->mats[i].p, mats9 + 9 * i + 1, 8 * sizeof(float));
  }
  s->num_mats = n_mats;
  s->lights = (light_t *) calloc(n_lights ? n_lights : 1, sizeof(light_t));
  for (uint32_t i = 0; i < n_lights; i++) s->lights[i] = light_from_args(lights16 + 16 * i);
  s->num_lights = n_lights;

  bnode_t *root = construct_bvh(v, 0, n, leaf_size);
  s->num_prims = n;
  s->num_nodes = count_nodes(root);
  s->depth = max_depth(root, 0);
  s->prims = (prim_t *) calloc(n ? n : 1, sizeof(prim_t));
  s->nodes = (node_t *) calloc(s->num_nodes, sizeof(node_t));
  uint32_t ps = 0, ns = 0;
  flatten(root, v, s->prims, s->nodes, &ps, &ns);
  free_tree(root);
  free(v);
  free(store);
  return s;
}
void orc_scene_destroy(void *h) {
  scene_t *s = (scene_t *) h;
  if (!s) return;
  free(s->prims); free(s->nodes); free(s->mats); free(s->lights); free(s);
}
void orc_scene_info(void *h, uint32_t *n_nodes, uint32_t *depth) {
  scene_t *s = (scene_t *) h;
  *n_nodes = s->num_nodes; *depth = s->depth;
}
void orc_scene_bvh(void *h, uint32_t *nodes_out, uint32_t *flat_orig) {
  scene_t *s = (scene_t *) h;
  memcpy(nodes_out, s->nodes, (size_t) s->num_nodes * 32);
  for (uint32_t i = 0; i < s->num_prims; i++) flat_orig[i] = s->prims[i].orig;
}
void orc_set_rng(void *h, int mode, uint32_t seed) { scene_t *s = (scene_t *) h; s->rng_mode = mode; s->seed = seed; }

void orc_set_camera(void *h, const float *from, const float *at, const float *up, float hfov, float vfov,
                    float nclip, float fclip, float aperture, float focal) {              /* camera.cuh:19-36 */
  scene_t *s = (scene_t *) h;
  v3 f = P3(from), a = P3(at), u = P3(up);
  if (focal == 0) focal = vnorm(vsub(f, a));                                              /* renderer.cuh:97-99 */
  s->nClip = nclip; s->fClip = fclip; s->hFov = hfov; s->vFov = vfov;
  s->cam_origin = f; s->aperture = aperture; s->focal_distance = focal;
  v3 dirToCamera = vsub(f, a);
  v3 sx = vunit(vcross(u, dirToCamera));
  v3 sy = vunit(vcross(dirToCamera, sx));
  s->c2w.c[0] = sx; s->c2w.c[1] = sy; s->c2w.c[2] = vunit(dirToCamera);
}
void orc_get_camera(void *h, float *out18) {
  scene_t *s = (scene_t *) h;
  memcpy(out18, &s->cam_origin, 12);
  memcpy(out18 + 3, &s->c2w, 36);
  out18[12] = s->hFov; out18[13] = s->vFov; out18[14] = s->nClip; out18[15] = s->fClip;
  out18[16] = s->aperture; out18[17] = s->focal_distance;
}

typedef struct {
  const scene_t *s; params_t prm; uint32_t w, h; int32_t *rgb; float *accum; uint32_t sample_begin;
  int tid, nthreads; counters_t cnt;
} job_t;

static void *render_rows(void *arg) {
  job_t *j = (job_t *) arg;
  memset(&j->cnt, 0, sizeof j->cnt);
  for (uint32_t y = (uint32_t) j->tid; y < j->h; y += (uint32_t) j->nthreads)
    for (uint32_t x = 0; x < j->w; x++) {
      size_t i = (size_t) y * j->w + x;
      rng_t seed = pixel_rng(j->s, y * j->w + x);                        /* renderer.cuh:28-29 */
      fill_color(x, y, j->w, j->h, j->s, &j->prm, &seed, j->rgb ? j->rgb + 3 * i : NULL,
                 j->accum ? j->accum + 4 * i : NULL, j->sample_begin, &j->cnt);
    }
  return NULL;
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* params5: spp spl batch depth leaf.  counters_out (4 x u64): closest rays, shadow rays, box tests, tri tests.
 * Returns the wall time of the pixel loop in seconds. */
double orc_render(void *h, const uint32_t *params5, float tol, uint32_t w, uint32_t hgt, int32_t *rgb_out,
                  float *accum_out, uint32_t sample_begin, int nthreads, uint64_t *counters_out) {
  scene_t *s = (scene_t *) h;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  job_t jobs[256];
  pthread_t th[256];
  double t0 = now_s();
  for (int t = 0; t < nthreads; t++) {
    job_t *j = &jobs[t];
    j->s = s;
    j->prm.spp = params5[0]; j->prm.spl = params5[1]; j->prm.batch = params5[2]; j->prm.tol = tol;
    j->prm.depth = params5[3]; j->prm.leaf = params5[4];
    j->w = w; j->h = hgt; j->rgb = rgb_out; j->accum = accum_out; j->sample_begin = sample_begin;
    j->tid = t; j->nthreads = nthreads;
    if (nthreads == 1) render_rows(j);
    else pthread_create(&th[t], NULL, render_rows, j);
  }
  if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  double t1 = now_s();
  if (counters_out) {
    memset(counters_out, 0, 4 * sizeof(uint64_t));
    for (int t = 0; t < nthreads; t++) {
      counters_out[0] += jobs[t].cnt.closest; counters_out[1] += jobs[t].cnt.shadow;
      counters_out[2] += jobs[t].cnt.box; counters_out[3] += jobs[t].cnt.tri;
    }
  }
  return t1 - t0;
}

static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static void store_ray(const ray_t *ray, float *r) {
  r[0] = ray->origin.x; r[1] = ray->origin.y; r[2] = ray->origin.z;
  r[3] = ray->direction.x; r[4] = ray->direction.y; r[5] = ray->direction.z;
  r[6] = ray->inv_d.x; r[7] = ray->inv_d.y; r[8] = ray->inv_d.z;
  r[9] = ray->min_t; r[10] = ray->max_t;
}

void orc_primary_hits(void *h, uint32_t w, uint32_t hgt, int32_t *idx_out, uint32_t *tbits_out, float *ray_out) {
  scene_t *s = (scene_t *) h;
  for (uint32_t y = 0; y < hgt; y++)
    for (uint32_t x = 0; x < w; x++) {
      rng_t seed = pixel_rng(s, y * w + x);
      if (seed.mode == 1) { seed.state = 0; seed.sample = 0; }
      v2 g = sample_grid2(&seed);
      float px = (float) x + g.x, py = (float) y + g.y;
      ray_t ray = camera_generate_ray(s, px / (float) w, py / (float) hgt);
      isect_raw_t raw;
      raw.hit = 0; raw.prim = 0; raw.normal = V(0, 0, 0);
      int hit = bvh_intersect(s, &ray, &raw, NULL, NULL);
      size_t i = (size_t) y * w + x;
      idx_out[i] = hit ? (int32_t) raw.prim : -1;
      tbits_out[i] = fbits(ray.max_t);
      if (ray_out) store_ray(&ray, ray_out + 11 * i);
    }
}

static ray_t ray_from8(const float *r) {
  ray_t ray = make_ray(P3(r), P3(r + 3));
  ray.min_t = r[6]; ray.max_t = r[7];
  return ray;
}

void orc_trace(void *h, const float *rays, uint32_t n, int anyhit, int32_t *idx_out, uint32_t *tbits_out,
               float *nrm_out) {
  scene_t *s = (scene_t *) h;
  for (uint32_t i = 0; i < n; i++) {
    ray_t ray = ray_from8(rays + 8 * i);
    if (anyhit) {
      int hit = bvh_has_intersection(s, &ray, NULL, NULL);
      idx_out[i] = hit ? 1 : -1;
      tbits_out[i] = fbits(ray.max_t);
    } else {
      isect_raw_t raw;
      raw.hit = 0; raw.prim = 0; raw.normal = V(0, 0, 0);
      int hit = bvh_intersect(s, &ray, &raw, NULL, NULL);
      idx_out[i] = hit ? (int32_t) raw.prim : -1;
      tbits_out[i] = fbits(ray.max_t);
      if (nrm_out) {
        nrm_out[3 * i] = hit ? raw.normal.x : 0; nrm_out[3 * i + 1] = hit ? raw.normal.y : 0;
        nrm_out[3 * i + 2] = hit ? raw.normal.z : 0;
      }
    }
  }
}

void orc_radiance(void *h, const uint32_t *params5, uint32_t w, uint32_t hgt, uint32_t seed_xor, float *rgb_out,
                  int nthreads) {
  scene_t *s = (scene_t *) h;
  scene_t tmp = *s;
  tmp.seed = seed_xor;
  uint32_t prm[5] = {params5[0], params5[1], params5[0] + 1, params5[3], params5[4]};   /* batch > spp: no stop */
  float *acc = (float *) calloc((size_t) w * hgt * 4, sizeof(float));
  orc_render(&tmp, prm, 0.f, w, hgt, NULL, acc, 0, nthreads, NULL);
  for (size_t i = 0; i < (size_t) w * hgt; i++) {
    v3 c = vdivassign(V(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2]), (float) params5[0]);
    rgb_out[3 * i] = c.x; rgb_out[3 * i + 1] = c.y; rgb_out[3 * i + 2] = c.z;
  }
  free(acc);
}

/* ---- unit probes (same signatures as ref_harness.cpp) ---------------------------------------- */
float orc_random_float(uint32_t *seed) {
  rng_t r; r.mode = 0; r.state = *seed; r.key = 0; r.sample = 0;
  float f = random_float(&r);
  *seed = r.state;
  return f;
}
uint32_t orc_ctr_u32(uint32_t key, uint32_t sample, uint32_t n) {
  uint32_t o0, o1;
  philox2x32_10(n >> 1, sample, key, &o0, &o1);
  return (n & 1) ? o1 : o0;
}
int orc_bbox_intersect(const float *minp, const float *maxp, const float *ray8) {
  ray_t ray = ray_from8(ray8);
  return bbox_intersect(P3(minp), P3(maxp), &ray);
}
int orc_tri_intersect(const float *tri, const float *ray8, int anyhit, uint32_t *tbits, float *nrm) {
  ray_t ray = ray_from8(ray8);
  prim_t p;
  memset(&p, 0, sizeof p);
  p.type = PRIM_TRIANGLE;
  memcpy(p.d, tri, 18 * sizeof(float));
  isect_raw_t raw;
  raw.hit = 0; raw.prim = 0; raw.normal = V(0, 0, 0);
  int hit = anyhit ? prim_has_intersection(&p, &ray) : prim_intersect(&p, 0, &ray, &raw);
  *tbits = fbits(ray.max_t);
  nrm[0] = raw.normal.x; nrm[1] = raw.normal.y; nrm[2] = raw.normal.z;
  return hit;
}
void orc_make_coord_space(const float *n, float *o2w9) {
  m3 m;
  make_coord_space(&m, P3(n));
  memcpy(o2w9, &m, 36);
}
void orc_isect_compute(const float *ray8, float max_t, const float *normal, float *out27) {
  ray_t ray = ray_from8(ray8);
  ray.max_t = max_t;
  isect_t is;
  is.prim = 0;
  is.normal = P3(normal);
  isect_compute(&is, &ray);
  memcpy(out27, &is.normal, 12);
  memcpy(out27 + 3, &is.o2w, 36);
  memcpy(out27 + 12, &is.w2o, 36);
  memcpy(out27 + 21, &is.hit_point, 12);
  memcpy(out27 + 24, &is.o_out, 12);
}
static bsdf_t bsdf_from9(const float *m) { bsdf_t b; b.type = (int) m[0]; memcpy(b.p, m + 1, 32); return b; }
void orc_bsdf_f(const float *mat9, const float *wo, const float *wi, float *out3) {
  bsdf_t b = bsdf_from9(mat9);
  v3 f = bsdf_f(&b, P3(wo), P3(wi));
  out3[0] = f.x; out3[1] = f.y; out3[2] = f.z;
}
int orc_bsdf_sample(const float *mat9, const float *wo, uint32_t *seed, float *out7) {
  bsdf_t b = bsdf_from9(mat9);
  rng_t r; r.mode = 0; r.state = *seed; r.key = 0; r.sample = 0;
  v3 mask = V(0, 0, 0);
  float pdf = 0;
  v3 wi = bsdf_sample(&b, P3(wo), &mask, &pdf, &r);
  *seed = r.state;
  out7[0] = wi.x; out7[1] = wi.y; out7[2] = wi.z; out7[3] = mask.x; out7[4] = mask.y; out7[5] = mask.z; out7[6] = pdf;
  return bsdf_is_delta(&b);
}
void orc_light_sample(const float *light16, const float *p, uint32_t *seed, float *out8) {
  light_t l = light_from_args(light16);
  rng_t r; r.mode = 0; r.state = *seed; r.key = 0; r.sample = 0;
  v3 wi = V(0, 0, 0);
  float dist = 0, pdf = 0;
  v3 L = light_sample(&l, P3(p), &wi, &dist, &pdf, &r);
  *seed = r.state;
  out8[0] = L.x; out8[1] = L.y; out8[2] = L.z; out8[3] = wi.x; out8[4] = wi.y; out8[5] = wi.z; out8[6] = dist; out8[7] = pdf;
}
void orc_light_record(const float *light16, uint32_t *out17) {
  light_t l = light_from_args(light16);
  memcpy(out17, &l, 68);
}
void orc_update_pixel(const float *c, int32_t *out3) { update_pixel(P3(c), out3); }
void orc_generate_ray(void *h, float x, float y, uint32_t *seed, float *ray_out) {
  scene_t *s = (scene_t *) h;
  rng_t r; r.mode = 0; r.state = *seed; r.key = 0; r.sample = 0;
  ray_t ray = (s->aperture < EPS_F) ? camera_generate_ray(s, x, y) : camera_generate_ray_thin_lens(s, x, y, &r);
  *seed = r.state;
  store_ray(&ray, ray_out);
}
