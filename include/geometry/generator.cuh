// Forwarding header at the reference's include path (reference: include/geometry/generator.cuh): Generator::room and
// Generator::from_obj (own OBJ / MTL reader) are part of raytracer_amd/pathtracer.hpp.
#pragma once
#include "../raytracer_amd/pathtracer.hpp"
