// PtVectorTypes.h -- the float3 the reference's public interface is written in
// (Camera::getEyeRayBasis(float3*, ...), Sphere::pos ...).  Host-only POD; define
// PT_HAVE_FLOAT3 before including these headers if another header already provides float3.
#ifndef PT_VECTOR_TYPES_H
#define PT_VECTOR_TYPES_H
#ifndef PT_HAVE_FLOAT3
struct float3 {
  float x, y, z;
};
static inline float3 make_float3(float x, float y, float z) {
  float3 r = {x, y, z};
  return r;
}
#endif
#endif
