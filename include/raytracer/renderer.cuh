// Forwarding header at the reference's include path (reference: include/raytracer/renderer.cuh): an application written
// against the reference - `#include "raytracer/renderer.cuh"`, then PathTracer / Parameters / the factories - compiles
// unchanged against this tree and renders through libmirt.so on an MI355X.  The API itself is in raytracer_amd/pathtracer.hpp.
#pragma once
#include "../raytracer_amd/pathtracer.hpp"
