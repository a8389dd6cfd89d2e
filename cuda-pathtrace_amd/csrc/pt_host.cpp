// pt_host.cpp -- host-side inputs of the path: scene tables and the camera's eye-ray basis.
// Pure C++ (no HIP): usable on a machine without a GPU.
#include <cmath>
#include <cstring>

#include "../../include/ptcore.h"
#include "pt_internal.h"

namespace {

// include/Scene.h:26-34, values verbatim (smallpt Cornell box).
const pt_sphere kCornell[9] = {
    {1e5f, {1e5f + 1.0f, 40.8f, 81.6f}, {0.0f, 0.0f, 0.0f}, {0.75f, 0.25f, 0.25f}},     // Left
    {1e5f, {-1e5f + 99.0f, 40.8f, 81.6f}, {0.0f, 0.0f, 0.0f}, {.25f, .25f, .75f}},      // Right
    {1e5f, {50.0f, 40.8f, 1e5f}, {0.0f, 0.0f, 0.0f}, {.75f, .75f, .75f}},               // Back
    {1e5f, {50.0f, 40.8f, -1e5f + 600.0f}, {0.0f, 0.0f, 0.0f}, {1.00f, 1.00f, 1.00f}},  // Front
    {1e5f, {50.0f, 1e5f, 81.6f}, {0.0f, 0.0f, 0.0f}, {.75f, .75f, .75f}},               // Bottom
    {1e5f, {50.0f, -1e5f + 81.6f, 81.6f}, {0.0f, 0.0f, 0.0f}, {.75f, .75f, .75f}},      // Top
    {16.5f, {27.0f, 16.5f, 47.0f}, {0.0f, 0.0f, 0.0f}, {1.0f, 1.0f, 1.0f}},             // small sphere 1
    {16.5f, {73.0f, 16.5f, 78.0f}, {0.0f, 0.0f, 0.0f}, {1.0f, 1.0f, 1.0f}},             // small sphere 2
    {600.0f, {50.0f, 681.6f - .78f, 81.6f}, {4.0f, 3.6f, 3.2f}, {0.0f, 0.0f, 0.0f}}     // Light
};

struct SplitMix64 {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  // uniform in [lo, hi), 24 random bits so the float is exact
  float uniform(float lo, float hi) { return lo + (hi - lo) * ((float)(next() >> 40) * (1.0f / 16777216.0f)); }
};

// ---- minimal float vector / matrix algebra in glm 0.9.8's operation order -------------------
struct V3 {
  float x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline V3 normalize(V3 v) {
  float k = 1.0f / std::sqrt(dot(v, v));  // glm::inversesqrt
  return {v.x * k, v.y * k, v.z * k};
}

struct M4 {
  float c[4][4];  // c[column][row]
};

M4 mul(const M4& a, const M4& b) {
  M4 r;
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++)
      r.c[j][i] = a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1] + a.c[2][i] * b.c[j][2] + a.c[3][i] * b.c[j][3];
  return r;
}

// 2x2 minor of rows (r0,r1) taken from columns (c0,c1)
inline float minor2(const M4& m, int c0, int c1, int r0, int r1) { return m.c[c0][r0] * m.c[c1][r1] - m.c[c1][r0] * m.c[c0][r1]; }

M4 inverse(const M4& m) {
  // glm::inverse(mat4): cofactors from 2x2 minors of the last three columns
  const float k00 = minor2(m, 2, 3, 2, 3), k02 = minor2(m, 1, 3, 2, 3), k03 = minor2(m, 1, 2, 2, 3);
  const float k04 = minor2(m, 2, 3, 1, 3), k06 = minor2(m, 1, 3, 1, 3), k07 = minor2(m, 1, 2, 1, 3);
  const float k08 = minor2(m, 2, 3, 1, 2), k10 = minor2(m, 1, 3, 1, 2), k11 = minor2(m, 1, 2, 1, 2);
  const float k12 = minor2(m, 2, 3, 0, 3), k14 = minor2(m, 1, 3, 0, 3), k15 = minor2(m, 1, 2, 0, 3);
  const float k16 = minor2(m, 2, 3, 0, 2), k18 = minor2(m, 1, 3, 0, 2), k19 = minor2(m, 1, 2, 0, 2);
  const float k20 = minor2(m, 2, 3, 0, 1), k22 = minor2(m, 1, 3, 0, 1), k23 = minor2(m, 1, 2, 0, 1);
  const float F[6][4] = {{k00, k00, k02, k03}, {k04, k04, k06, k07}, {k08, k08, k10, k11},
                         {k12, k12, k14, k15}, {k16, k16, k18, k19}, {k20, k20, k22, k23}};
  float V[4][4];
  for (int r = 0; r < 4; r++) {
    V[r][0] = m.c[1][r];
    V[r][1] = V[r][2] = V[r][3] = m.c[0][r];
  }
  M4 inv;
  for (int i = 0; i < 4; i++) {
    const float sa = (i & 1) ? -1.0f : 1.0f, sb = -sa;
    inv.c[0][i] = (V[1][i] * F[0][i] - V[2][i] * F[1][i] + V[3][i] * F[2][i]) * sa;
    inv.c[1][i] = (V[0][i] * F[0][i] - V[2][i] * F[3][i] + V[3][i] * F[4][i]) * sb;
    inv.c[2][i] = (V[0][i] * F[1][i] - V[1][i] * F[3][i] + V[3][i] * F[5][i]) * sa;
    inv.c[3][i] = (V[0][i] * F[2][i] - V[1][i] * F[4][i] + V[2][i] * F[5][i]) * sb;
  }
  const float det = (m.c[0][0] * inv.c[0][0] + m.c[0][1] * inv.c[1][0]) + (m.c[0][2] * inv.c[2][0] + m.c[0][3] * inv.c[3][0]);
  const float ood = 1.0f / det;
  for (auto& col : inv.c)
    for (float& e : col) e *= ood;
  return inv;
}

}  // namespace

extern "C" int pt_scene_cornell(pt_sphere out[9]) {
  if (!out) return pt_fail(PT_EINVAL, "pt_scene_cornell: out is NULL");
  std::memcpy(out, kCornell, sizeof(kCornell));
  return PT_OK;
}

extern "C" int pt_scene_random(int n, uint64_t seed, int with_walls, pt_sphere* out) {
  if (!out || n < 0) return pt_fail(PT_EINVAL, "pt_scene_random: bad arguments");
  if (with_walls && n < 7) return pt_fail(PT_EINVAL, "pt_scene_random: with_walls needs n >= 7");
  int i = 0;
  if (with_walls) {
    for (; i < 6; i++) out[i] = kCornell[i];
    out[i++] = kCornell[8];
  }
  SplitMix64 g{seed ^ 0x5EEDull};
  for (; i < n; i++) {
    pt_sphere s;
    s.pos[0] = g.uniform(1.0f, 99.0f);
    s.pos[1] = g.uniform(0.0f, 81.6f);
    s.pos[2] = g.uniform(0.0f, 170.0f);
    s.radius = g.uniform(0.5f, 3.0f);
    for (float& c : s.color) c = g.uniform(0.2f, 0.9f);
    const bool emissive = g.uniform(0.0f, 1.0f) < 0.01f;
    s.emission[0] = emissive ? 4.0f : 0.0f;
    s.emission[1] = emissive ? 3.6f : 0.0f;
    s.emission[2] = emissive ? 3.2f : 0.0f;
    out[i] = s;
  }
  return PT_OK;
}

extern "C" int pt_camera_basis(const float pos[3], float yaw_deg, float pitch_deg, int width, int height,
                               float basis_out[12]) {
  const float world_up[3] = {0.0f, 1.0f, 0.0f};  // Camera.h:58
  return pt_camera_basis_up(pos, yaw_deg, pitch_deg, world_up, width, height, basis_out);
}

extern "C" int pt_camera_basis_up(const float pos[3], float yaw_deg, float pitch_deg, const float world_up[3], int width,
                                  int height, float basis_out[12]) {
  if (!pos || !world_up || !basis_out || width <= 0 || height <= 0) return pt_fail(PT_EINVAL, "pt_camera_basis: bad arguments");
  const float rad = 0.01745329251994329576923690768489f;  // glm::radians
  // Camera::updateCameraVectors, include/Camera.h:153-164
  const float yaw = yaw_deg * rad, pitch = pitch_deg * rad;
  const V3 front = normalize(V3{std::cos(yaw) * std::cos(pitch), std::sin(pitch), std::sin(yaw) * std::cos(pitch)});
  const V3 right = normalize(cross(front, V3{world_up[0], world_up[1], world_up[2]}));
  const V3 up = normalize(cross(right, front));
  const V3 eye{pos[0], pos[1], pos[2]};
  // Camera::GetViewMatrix -> glm::lookAt (right-handed), include/Camera.h:73-76
  const V3 f = normalize((eye + front) - eye);
  const V3 s = normalize(cross(f, up));
  const V3 u = cross(s, f);
  M4 view{};
  view.c[0][0] = s.x; view.c[1][0] = s.y; view.c[2][0] = s.z;
  view.c[0][1] = u.x; view.c[1][1] = u.y; view.c[2][1] = u.z;
  view.c[0][2] = -f.x; view.c[1][2] = -f.y; view.c[2][2] = -f.z;
  view.c[3][0] = -dot(s, eye); view.c[3][1] = -dot(u, eye); view.c[3][2] = dot(f, eye);
  view.c[3][3] = 1.0f;
  // glm::perspective(radians(45), w/(float)h, 0.01f, 1000.0f), include/Camera.h:130
  const float aspect = (float)width / (float)height, zn = 0.01f, zf = 1000.0f;
  const float th = std::tan((45.0f * rad) / 2.0f);
  M4 proj{};
  proj.c[0][0] = 1.0f / (aspect * th);
  proj.c[1][1] = 1.0f / th;
  proj.c[2][2] = -(zf + zn) / (zf - zn);
  proj.c[2][3] = -1.0f;
  proj.c[3][2] = -(2.0f * zf * zn) / (zf - zn);
  const M4 inv = inverse(mul(proj, view));  // include/Camera.h:131
  // include/Camera.h:132-148
  const float ndc[4][2] = {{-1, -1}, {+1, -1}, {-1, +1}, {+1, +1}};
  for (int k = 0; k < 4; k++) {
    float r[4];
    for (int i = 0; i < 4; i++)
      r[i] = (inv.c[0][i] * ndc[k][0] + inv.c[1][i] * ndc[k][1]) + (inv.c[2][i] * 0.0f + inv.c[3][i] * 1.0f);
    basis_out[3 * k + 0] = r[0] / r[3] - eye.x;
    basis_out[3 * k + 1] = r[1] / r[3] - eye.y;
    basis_out[3 * k + 2] = r[2] / r[3] - eye.z;
  }
  return PT_OK;
}
