// Camera.h -- look-alike of include/Camera.h without glm/GLEW: the learnopengl fly camera
// reduced to what the hot path consumes (Position + getEyeRayBasis) plus the movement methods
// a scripted fly-through needs.  The matrix pipeline (lookAt, perspective, inverse) lives in
// libptcore's pt_camera_basis, restated in glm 0.9.8's float32 operation order.
#ifndef CAMERA_H
#define CAMERA_H
#include <math.h>

#include "HipErrorCheck.h"
#include "PtVectorTypes.h"

enum Camera_Movement { FORWARD, BACKWARD, LEFT, RIGHT };  // Camera.h:20-25

const float YAW = -90.0f;         // Camera.h:28-32
const float PITCH = 0.0f;
const float SPEED = 50.0f;
const float SENSITIVTY = 1.25f;
const float ZOOM = 45.0f;

class Camera {
 public:
  float3 Position;  // glm::vec3 in the reference; memcpy'd as a float3 (Renderer.h:60)
  float3 Front, Up, Right, WorldUp;
  float Yaw, Pitch;
  float MovementSpeed, MouseSensitivity, Zoom;

  // Camera.h:54-61 (glm::vec3 position replaced by its three floats)
  Camera(float posX = 0.0f, float posY = 0.0f, float posZ = 0.0f, float yaw = YAW, float pitch = PITCH)
      : MovementSpeed(SPEED), MouseSensitivity(SENSITIVTY), Zoom(ZOOM) {
    Position = make_float3(posX, posY, posZ);
    WorldUp = make_float3(0.0f, 1.0f, 0.0f);
    Yaw = yaw;
    Pitch = pitch;
    updateCameraVectors();
  }

  void ProcessKeyboard(Camera_Movement direction, float deltaTime) {  // Camera.h:79-90
    float v = MovementSpeed * deltaTime;
    if (direction == FORWARD) Position = make_float3(Position.x + Front.x * v, Position.y + Front.y * v, Position.z + Front.z * v);
    if (direction == BACKWARD) Position = make_float3(Position.x - Front.x * v, Position.y - Front.y * v, Position.z - Front.z * v);
    if (direction == LEFT) Position = make_float3(Position.x - Right.x * v, Position.y - Right.y * v, Position.z - Right.z * v);
    if (direction == RIGHT) Position = make_float3(Position.x + Right.x * v, Position.y + Right.y * v, Position.z + Right.z * v);
  }

  void ProcessMouseMovement(float xoffset, float yoffset, bool constrainPitch = true) {  // Camera.h:93-112
    Yaw += xoffset * MouseSensitivity;
    Pitch += yoffset * MouseSensitivity;
    if (constrainPitch) {
      if (Pitch > 89.0f) Pitch = 89.0f;
      if (Pitch < -89.0f) Pitch = -89.0f;
    }
    updateCameraVectors();
  }

  // Camera.h:125-149: four un-normalised corner directions, order (-1,-1) (+1,-1) (-1,+1) (+1,+1).
  void getEyeRayBasis(float3* output, int w, int h) const {
    float pos[3] = {Position.x, Position.y, Position.z};
    float out[12];
    gpuErrchk(pt_camera_basis(pos, Yaw, Pitch, w, h, out));
    for (int k = 0; k < 4; k++) output[k] = make_float3(out[3 * k], out[3 * k + 1], out[3 * k + 2]);
  }

 private:
  static float3 norm(float3 v) {
    float k = 1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return make_float3(v.x * k, v.y * k, v.z * k);
  }
  static float3 cross(float3 a, float3 b) {
    return make_float3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
  }
  void updateCameraVectors() {  // Camera.h:153-164
    const float rad = 0.01745329251994329576923690768489f;
    float3 front = make_float3(cosf(Yaw * rad) * cosf(Pitch * rad), sinf(Pitch * rad), sinf(Yaw * rad) * cosf(Pitch * rad));
    Front = norm(front);
    Right = norm(cross(Front, WorldUp));
    Up = norm(cross(Right, Front));
  }
};
#endif
