// Forwarding header at the reference's include path (reference: include/physics/physics.cuh, physics/particle.cuh):
// PhysicsSystem / Particle / kelvinToRGB of the particle demo live in raytracer_amd/physics.hpp.
#pragma once
#include "../raytracer_amd/physics.hpp"
