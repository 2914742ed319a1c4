// TEST INFRASTRUCTURE ONLY.  One program, two builds, same output expected bit for bit:
//   -DUSE_REFERENCE -I /root/reference/include   the reference's own Vector3f / Vector2f / Matrix3f (read where they lie)
//   -I include                                   the mirror of the kept host API (include/raytracer_amd/linalg.hpp)
// It evaluates every operator of the three types on seeded operands and - because the reference's particle header does not
// compile under g++ (physics/particle.cuh:77) and so cannot be run - the particle demo's update expressions written out
// with each build's own Vector3f (particle.cuh:60-66, :80-85): what the mirror's Particle::step must reproduce.
// Not probed: outer(), which in the reference writes nine doubles into nine floats (Matrix3f.cuh:261-276: stack smash).
// Output: one line per probe, hex bit patterns.  oracle/Makefile writes the reference build's output to
// tests/golden/linalg_probe.txt; tests/test_host.py compares the mirror build (and physics.hpp's Particle) with it.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#ifdef USE_REFERENCE
#include "raytracer/linalg/Vector3f.cuh"
#include "raytracer/linalg/Vector2f.cuh"
#include "raytracer/linalg/Matrix3f.cuh"
#else
#include "raytracer_amd/linalg.hpp"
#ifdef WITH_PARTICLE
#include "raytracer_amd/physics.hpp"
#endif
#endif

static uint32_t rng_state = 0x9E3779B9u;
static float rnd(float lo, float hi) {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5;
  return lo + (hi - lo) * (float) (rng_state >> 8) * (1.0f / 16777216.0f);
}
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static void out(const char *tag, int i, float v) { printf("%s %d %08x\n", tag, i, bits(v)); }
static void out3(const char *tag, int i, const Vector3f &v) { printf("%s %d %08x %08x %08x\n", tag, i, bits(v.x), bits(v.y), bits(v.z)); }
static void out2(const char *tag, int i, const Vector2f &v) { printf("%s %d %08x %08x\n", tag, i, bits(v.x), bits(v.y)); }
static void outm(const char *tag, int i, const Matrix3f &m) {
  printf("%s %d", tag, i);
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) printf(" %08x", bits(m(r, c)));
  printf("\n");
}

int main() {
  for (int i = 0; i < 60; i++) {
    const float s = (i % 3 == 0) ? 1000.f : ((i % 3 == 1) ? 1.f : 1e-3f);
    Vector3f a(rnd(-s, s), rnd(-s, s), rnd(-s, s)), b(rnd(-s, s), rnd(-s, s), rnd(-s, s));
    const float c = rnd(-7.f, 7.f);
    out("dot", i, a.dot(b)); out3("cross", i, a.cross(b)); out3("neg", i, -a); out3("add", i, a + b); out3("sub", i, a - b);
    out3("mul", i, a * b); out3("divv", i, a / b); out3("muls", i, a * c); out3("divs", i, a / c); out3("smul", i, c * a);
    out3("sdiv", i, c / a); out3("rcp", i, a.rcp()); out("norm", i, a.norm()); out("norm2", i, a.norm2()); out3("unit", i, a.unit());
    out("illum", i, a.illum()); out("idx", i, a[1]); out("eq", i, (float) (a == b) + (float) (a == a));
    Vector3f t = a; t += b; out3("pluseq", i, t); t -= a; out3("minuseq", i, t); t *= c; out3("muleq", i, t); t /= c; out3("diveq", i, t);
    t = a; t.normalize(); out3("normalize", i, t);
    out3("splat", i, Vector3f(c));
    Vector2f p(rnd(-s, s), rnd(-s, s)), q(rnd(-s, s), rnd(-s, s));
    out("dot2", i, p.dot(q)); out("cross2", i, p.cross(q)); out2("neg2", i, -p); out2("add2", i, p + q); out2("sub2", i, p - q);
    out2("muls2", i, p * c); out2("divs2", i, p / c); out2("smul2", i, c * p); out("norm_2", i, p.norm()); out("norm2_2", i, p.norm2());
    out2("unit2", i, p.unit());
    Vector2f u = p; u += q; u -= p; u *= c; u /= c; out2("assign2", i, u);
    Matrix3f M(rnd(-2, 2), rnd(-2, 2), rnd(-2, 2), rnd(-2, 2), rnd(-2, 2), rnd(-2, 2), rnd(-2, 2), rnd(-2, 2), rnd(-2, 2));
    Matrix3f N = Matrix3f::crossProduct(a);
    outm("M", i, M); outm("T", i, M.T()); outm("inv", i, M.inv()); out("det", i, M.det()); out("mnorm", i, M.norm());
    out3("Mv", i, M * b); outm("MN", i, M * N); outm("Ms", i, M * c); outm("sM", i, c * M); outm("negM", i, -M); outm("subM", i, M - N);
    out3("col", i, M.column(2)); out3("colidx", i, M[1]); outm("ident", i, Matrix3f::identity());
    Matrix3f Z; Z.zero(c); Z += M; outm("zero_pluseq", i, Z);
    float rows[9]; for (float &r: rows) r = rnd(-1, 1);
    outm("rowctor", i, Matrix3f(rows));
  }
  // the particle demo's update (particle.cuh:60-66, :80-85), 40 steps of dt = 0.05 for 50 particles
  for (int k = 0; k < 50; k++) {
    Vector3f position(rnd(-1, 1), rnd(0, 2), rnd(-1, 1)), velocity(rnd(-30, 30), rnd(0, 60), rnd(-30, 30));
    const float mass = 1, radius = .2, dt = 0.05f;
#if !defined(USE_REFERENCE) && defined(WITH_PARTICLE)
    Particle pt(position, velocity, mass, radius, 3000.f, 2000.f);       // the mirror's class (include/raytracer_amd/physics.hpp)
    for (int step = 0; step < 40; step++) pt.step(dt);
    position = pt.position; velocity = pt.velocity;
#else
    for (int step = 0; step < 40; step++) {
      Vector3f drag = 0.45 * pow(radius, 2) * velocity * velocity.norm();
      Vector3f forces = Vector3f(0, 1, 0) * -9.8 * mass - drag;
      Vector3f acceleration = forces / mass;
      position += velocity * dt + acceleration * pow(dt, 2) / 2;
      velocity += acceleration * dt;
      if (position.y < 0) { position.y = 0; velocity.y = 0; }
    }
#endif
    out3("particle_pos", k, position); out3("particle_vel", k, velocity);
  }
  return 0;
}
