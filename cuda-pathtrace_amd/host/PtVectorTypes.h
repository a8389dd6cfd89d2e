// PtVectorTypes.h -- the vector types the reference's public interface is written in: CUDA's float3
// (Camera::getEyeRayBasis(float3*, ...), Sphere::pos ...) and glm::vec3 (Camera's constructor and public
// members, include/Camera.h:40-44,54; the call site src/main.cu:128).  Host-only PODs with the members and
// operators those uses need; define PT_HAVE_FLOAT3 / PT_HAVE_GLM before including these headers if the real
// vector_types.h / glm are present.
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
#ifndef PT_HAVE_GLM
namespace glm {
// three packed floats like glm::vec3: the reference memcpy's camera.Position into a float3 (Renderer.h:60)
struct vec3 {
  float x, y, z;
  vec3() : x(0.0f), y(0.0f), z(0.0f) {}
  vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
  explicit vec3(float s) : x(s), y(s), z(s) {}
  vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
  vec3& operator-=(const vec3& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
};
inline vec3 operator+(vec3 a, const vec3& b) { return a += b; }
inline vec3 operator-(vec3 a, const vec3& b) { return a -= b; }
inline vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, const vec3& a) { return a * s; }
}  // namespace glm
#endif
#endif
