// C++ mirror of the reference's particle demo (reference: include/physics/physics.cuh:10-44, physics/particle.cuh:9-97), the
// scene generator behind apps/app.cu: N emissive spheres thrown from a point, gravity + quadratic drag, black-body colour by
// temperature.  Host code only; it feeds PathTracer::set_scene (pathtracer.hpp) like any other primitive list.
// One deliberate divergence: particle.cuh:77 calls maxf(temp, 100) with an int, which does not compile on the reference's
// own g++ path (std::max<float, int>); here it is the float comparison the CUDA build performs.
#pragma once
#include <cmath>
#include <vector>

#include "pathtracer.hpp"

namespace mirt_physics {

// util/sampler.cuh:8-15
inline float random_float(uint *state) {
  uint x = *state;
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  *state = x;
  return float(x) / 4294967296.0f;                          // float(UINT_MAX) rounds to 2^32
}

// util/sampler.cuh:52-63
inline Vector3f sample_hemisphere(uint *seed) {
  const float PI_F = 3.14159265358979323f;
  float Xi1 = random_float(seed);
  float Xi2 = random_float(seed);
  float theta = acosf(Xi1);
  float phi = 2.0f * PI_F * Xi2;
  return {sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta)};
}

// scene/bsdf.cuh:9-28: columns x, y, z of the object-to-world frame around n
struct Frame {
  Vector3f x, y, z;
  Vector3f apply(const Vector3f &v) const { return x * v.x + y * v.y + z * v.z; }   // Matrix3f.cuh:243-247
};
inline Frame make_coord_space(const Vector3f &n) {
  Vector3f z = n, h = n;
  if (fabsf(h.x) <= fabsf(h.y) && fabsf(h.x) <= fabsf(h.z)) h.x = 1.0f;
  else if (fabsf(h.y) <= fabsf(h.x) && fabsf(h.y) <= fabsf(h.z)) h.y = 1.0f;
  else h.z = 1.0f;
  z = z.unit();
  Vector3f y = h.cross(z).unit();
  Vector3f x = z.cross(y).unit();
  return {x, y, z};
}

inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace mirt_physics

// particle.cuh:9-36 (double-typed literals and log / pow as there)
static Vector3f kelvinToRGB(float k) {
  float temp = k / 100;
  float red, green, blue;
  if (temp <= 66) {
    red = (float) (61 * log((double) temp));
    green = temp;
    green = (float) (99.4708025861 * log((double) green) - 161.1195681661);
    if (temp <= 19) {
      blue = 0;
    } else {
      blue = temp - 10;
      blue = (float) (138.5177312231 * log((double) blue) - 305.0447927307);
    }
  } else {
    red = temp - 60;
    red = (float) (329.698727446 * pow((double) red, -0.1332047592));
    green = temp - 60;
    green = (float) (288.1221695283 * pow((double) green, -0.0755148492));
    blue = 255;
  }
  using mirt_physics::clampf;
  return {clampf(red, 0.f, 255.f) / 255, clampf(green, 0.f, 255.f) / 255, clampf(blue, 0.f, 255.f) / 255};
}

class Particle {                                             // particle.cuh:39-97
public:
  Vector3f position, velocity;
  float mass, radius, temp, temp_decay;

  Particle(const Vector3f &starting_position, const Vector3f &starting_velocity, float starting_mass, float starting_radius,
           float starting_temp, float starting_temp_decay)
      : position(starting_position), velocity(starting_velocity), mass(starting_mass), radius(starting_radius),
        temp(starting_temp), temp_decay(starting_temp_decay) {}

  BSDF *new_bsdf() const { return new BSDF(BSDFFactory::createEmission(kelvinToRGB(temp))); }

  // The reference's Vector3f has only FLOAT scalar operators (Vector3f.cuh:101, :106, :180), so the double-typed factors of
  // these expressions - DRAG_COEF * pow(radius, 2), GRAVITY_ACCEL, pow(dt, 2) - are rounded to float where they meet a vector
  // and the products are formed in float, left to right.  The same expressions here, with those conversions spelled out.
  Vector3f get_drag() const {                                // particle.cuh:60-63
    const float c = (float) (0.45 * pow((double) radius, 2));
    return c * velocity * velocity.norm();
  }
  Vector3f get_forces(float) const {                         // particle.cuh:65-67
    return Vector3f(0, 1, 0) * (float) -9.8 * mass - get_drag();
  }
  void clamp_position() {
    if (position.y < 0) {
      position.y = 0;
      velocity.y = 0;
    }
  }
  void clamp_temperature() { temp = temp < 100.f ? 100.f : temp; }   // maxf(temp, 100), see the header comment
  void step_position(float dt) {
    Vector3f acceleration = get_forces(dt) / mass;           // particle.cuh:80-86
    position += velocity * dt + acceleration * (float) pow((double) dt, 2) / 2;
    velocity += acceleration * dt;
    clamp_position();
  }
  void step_temperature(float dt) {
    temp -= temp_decay * dt;
    clamp_temperature();
  }
  void step(float dt) {
    step_position(dt);
    step_temperature(dt);
  }
};

class PhysicsSystem {                                        // physics.cuh:10-44
public:
  std::vector<Particle *> particles;

  PhysicsSystem(uint num_particles, Vector3f starting_position, Vector3f up, float max_starting_velocity) {
    using namespace mirt_physics;
    const Frame o2w = make_coord_space(up);
    uint seed = 0x12345678;
    for (uint i = 0; i < 20; i++) (void) sample_hemisphere(&seed);           // the reference's unused "splotches" advance the RNG
    particles.reserve(num_particles);
    for (uint i = 0; i < num_particles; i++) {
      Vector3f direction = o2w.apply(sample_hemisphere(&seed));
      float speed = random_float(&seed) * max_starting_velocity;
      float temp = (2 * max_starting_velocity - speed) / (2 * max_starting_velocity) * 4000 + (random_float(&seed) * 500 - 250);
      // (the frame is applied a second time there: `o2w * direction * speed`, physics.cuh:28)
      particles.push_back(new Particle(starting_position, o2w.apply(direction) * speed, 1, .2f, temp, 2000));
    }
  }
  ~PhysicsSystem() { for (Particle *p: particles) delete p; }
  PhysicsSystem(const PhysicsSystem &) = delete;
  PhysicsSystem &operator=(const PhysicsSystem &) = delete;

  void to_primitives(std::vector<Primitive *> *primitives) {
    for (Particle *p: particles)
      primitives->push_back(new Primitive(PrimitiveFactory::createSphere(p->position, p->radius, p->new_bsdf())));
  }
  void step(float dt) { for (Particle *item: particles) item->step(dt); }
};
