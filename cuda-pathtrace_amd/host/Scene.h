// Scene.h -- look-alike of include/Scene.h (reference lines cited inline).
#ifndef SCENE_H
#define SCENE_H
#include <vector>

#include "HipErrorCheck.h"
#include "PtVectorTypes.h"

// A sphere object (Scene.h:7-14); same 40-byte layout as pt_sphere.
struct Sphere {
  float radius;
  float3 pos;
  // Material
  float3 emission;
  float3 color;
};
static_assert(sizeof(Sphere) == sizeof(pt_sphere), "Sphere must keep the reference's 40-byte layout");

// A scene which can be rendered (Scene.h:17-39): numObjects + device pointer, public fields.
class Scene {
 public:
  int numObjects;
  Sphere* objects;  // device memory

  // Scene.h:22-38: the 9-sphere smallpt Cornell box, uploaded in the constructor.
  Scene() {
    numObjects = 9;
    pt_sphere spheres[9];
    gpuErrchk(pt_scene_cornell(spheres));
    upload(spheres);
  }

  // Not in the reference (its scene is hard-coded): any host sphere list, e.g. the seeded
  // 1000-sphere scene of BASELINE.json config 4.
  explicit Scene(const std::vector<Sphere>& spheres) {
    numObjects = (int)spheres.size();
    upload(reinterpret_cast<const pt_sphere*>(spheres.data()));
  }

  static Scene Random(int n, unsigned long long seed, bool withWalls = true) {
    std::vector<Sphere> s(n);
    gpuErrchk(pt_scene_random(n, seed, withWalls ? 1 : 0, reinterpret_cast<pt_sphere*>(s.data())));
    return Scene(s);
  }

  // The reference never frees `objects` (Scene.h has no destructor and Scene is passed by
  // value into kernels); an explicit Free() is offered instead of changing copy semantics.
  void Free() {
    if (objects) gpuErrchk(pt_free(objects));
    objects = NULL;
  }

 private:
  void upload(const pt_sphere* host) {
    objects = NULL;
    void* d = NULL;
    gpuErrchk(pt_malloc(&d, (numObjects > 0 ? numObjects : 1) * sizeof(Sphere)));  // Scene.h:36
    if (numObjects > 0) gpuErrchk(pt_memcpy_h2d(d, host, numObjects * sizeof(Sphere)));  // Scene.h:37
    objects = static_cast<Sphere*>(d);
  }
};
#endif
