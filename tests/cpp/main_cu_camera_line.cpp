// The reference's own construction sequence, src/main.cu:125-131, compiled against the look-alike headers:
// Scene / Renderer / Camera(glm::vec3(...), yaw, pitch) / OutputBuffer must compile unchanged.
#include "cuda-pathtrace_amd/host/Camera.h"
#include "cuda-pathtrace_amd/host/OutputBuffer.h"
#include "cuda-pathtrace_amd/host/Renderer.h"
#include "cuda-pathtrace_amd/host/Scene.h"
int main() {
  int width = 32, height = 32, samplesPerPixel = 2, threadsPerBlock = 8;
  float cameraPos[3] = {50.0f, 52.0f, 295.6f}, cameraView[2] = {-90.0f, 0.0f};
  gpuErrchk(pt_set_device(0));
  // load scene and create renderer -- main.cu:125-128, verbatim
  Scene scene;
  Renderer renderer(width, height, samplesPerPixel, threadsPerBlock);
  Camera camera(glm::vec3(cameraPos[0], cameraPos[1], cameraPos[2]), cameraView[0], cameraView[1]);
  // allocate output buffer -- main.cu:131,138
  OutputBuffer d_buffer(width, height);
  d_buffer.AllocateGPU();
  float renderTime = renderer.Render(d_buffer, scene, camera);  // main.cu:182
  std::cout << "Render completed in " << renderTime << "ms (" << 1000.0f / renderTime << " fps)" << std::endl;  // :183
  Camera scalar(50.0f, 52.0f, 295.6f, 0.0f, 1.0f, 0.0f, -90.0f, 0.0f);  // Camera.h:63-70
  camera.ProcessKeyboard(FORWARD, 0.1f);
  camera.ProcessMouseScroll(50.0f);   // Camera.h:115-123: Zoom 45 -> clamped at 1
  scalar.ProcessMouseScroll(-3.0f);   // 45 -> 48 -> clamped at 45
  return (camera.Position.z < scalar.Position.z && camera.Zoom == 1.0f && scalar.Zoom == 45.0f) ? 0 : 1;
}
