// Camera.h -- look-alike of include/Camera.h without glm/GLEW: the learnopengl fly camera
// reduced to what the hot path consumes (Position + getEyeRayBasis) plus the movement methods
// a scripted fly-through needs.  Same two constructors and public members (glm::vec3 here is the
// three-float stand-in of PtVectorTypes.h), so main.cu:128 compiles against it unchanged.  The matrix pipeline (lookAt, perspective, inverse) lives in
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
  glm::vec3 Position;  // memcpy'd as a float3 by the reference (Renderer.h:60)
  glm::vec3 Front, Up, Right, WorldUp;
  float Yaw, Pitch;
  float MovementSpeed, MouseSensitivity, Zoom;

  // Constructor with vectors, Camera.h:54-61 -- the form src/main.cu:128 uses:
  //   Camera camera(glm::vec3(cameraPos[0], cameraPos[1], cameraPos[2]), cameraView[0], cameraView[1]);
  Camera(glm::vec3 position = glm::vec3(0.0f, 0.0f, 0.0f), float yaw = YAW, float pitch = PITCH)
      : Front(glm::vec3(0.0f, 0.0f, -1.0f)), MovementSpeed(SPEED), MouseSensitivity(SENSITIVTY), Zoom(ZOOM) {
    Position = position;
    WorldUp = glm::vec3(0.0f, 1.0f, 0.0f);
    Yaw = yaw;
    Pitch = pitch;
    updateCameraVectors();
  }
  // Constructor with scalar values, Camera.h:63-70
  Camera(float posX, float posY, float posZ, float upX, float upY, float upZ, float yaw, float pitch)
      : Front(glm::vec3(0.0f, 0.0f, -1.0f)), MovementSpeed(SPEED), MouseSensitivity(SENSITIVTY), Zoom(ZOOM) {
    Position = glm::vec3(posX, posY, posZ);
    WorldUp = glm::vec3(upX, upY, upZ);
    Yaw = yaw;
    Pitch = pitch;
    updateCameraVectors();
  }

  void ProcessKeyboard(Camera_Movement direction, float deltaTime) {  // Camera.h:79-90
    float velocity = MovementSpeed * deltaTime;
    if (direction == FORWARD) Position += Front * velocity;
    if (direction == BACKWARD) Position -= Front * velocity;
    if (direction == LEFT) Position -= Right * velocity;
    if (direction == RIGHT) Position += Right * velocity;
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

  // Camera.h:115-123.  (The field of view the rays are built with does not follow Zoom: getEyeRayBasis hard-codes 45 degrees,
  // Camera.h:130 -- scrolling changes the member and nothing else, here as there.)
  void ProcessMouseScroll(float yoffset) {
    if (Zoom >= 1.0f && Zoom <= 45.0f) Zoom -= yoffset;
    if (Zoom <= 1.0f) Zoom = 1.0f;
    if (Zoom >= 45.0f) Zoom = 45.0f;
  }

  // Camera.h:125-149: four un-normalised corner directions, order (-1,-1) (+1,-1) (-1,+1) (+1,+1).
  void getEyeRayBasis(float3* output, int w, int h) const {
    float pos[3] = {Position.x, Position.y, Position.z}, up[3] = {WorldUp.x, WorldUp.y, WorldUp.z};
    float out[12];
    gpuErrchk(pt_camera_basis_up(pos, Yaw, Pitch, up, w, h, out));
    for (int k = 0; k < 4; k++) output[k] = make_float3(out[3 * k], out[3 * k + 1], out[3 * k + 2]);
  }

 private:
  static glm::vec3 norm(glm::vec3 v) {
    float k = 1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return glm::vec3(v.x * k, v.y * k, v.z * k);
  }
  static glm::vec3 cross(glm::vec3 a, glm::vec3 b) {
    return glm::vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
  }
  void updateCameraVectors() {  // Camera.h:153-164
    const float rad = 0.01745329251994329576923690768489f;
    glm::vec3 front = glm::vec3(cosf(Yaw * rad) * cosf(Pitch * rad), sinf(Pitch * rad), sinf(Yaw * rad) * cosf(Pitch * rad));
    Front = norm(front);
    Right = norm(cross(Front, WorldUp));
    Up = norm(cross(Right, Front));
  }
};
#endif
