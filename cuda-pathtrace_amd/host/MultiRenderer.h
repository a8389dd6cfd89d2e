// MultiRenderer.h -- Renderer's interface (include/Renderer.h:23,48,55) over several GPUs of one node.
// No reference counterpart: the reference renders on the one device picked by cudaSetDevice (src/main.cu:86).
// Same constructor arguments plus the device list, same Render(d_buffer, scene, camera): the frame and the
// scene live on devices[0], exactly where a single-GPU caller has them; the row tiling, the per-device host
// threads and the RCCL exchange are inside libptcore (pt_mgpu_*, include/ptcore.h).
#ifndef MULTIRENDERER_H
#define MULTIRENDERER_H
#include <string>
#include <vector>

#include "Camera.h"
#include "HipErrorCheck.h"
#include "pathtrace.h"

class MultiRenderer {
 private:
  int width, height, samplesPerPixel;
  pt_mgpu* impl;

 public:
  MultiRenderer(const std::vector<int>& devices, int width, int height, int samplesPerPixel, int numThreads,
                const pt_renderer_opts* opts = NULL) {
    this->width = width;
    this->height = height;
    this->samplesPerPixel = samplesPerPixel;
    impl = NULL;
    gpuErrchk(pt_mgpu_create((int)devices.size(), devices.data(), width, height, samplesPerPixel, numThreads, opts, NULL, &impl));
  }
  MultiRenderer(const MultiRenderer&) = delete;
  MultiRenderer& operator=(const MultiRenderer&) = delete;
  ~MultiRenderer() { (void)pt_mgpu_destroy(impl); }

  // Synchronous like Renderer::Render; returns END-TO-END milliseconds (slowest tile + exchange).
  float Render(OutputBuffer d_buffer, const Scene& d_scene, const Camera& camera) {
    float3 eyeRayBasis[4];
    camera.getEyeRayBasis(eyeRayBasis, width, height);
    float basis[12], eye[3] = {camera.Position.x, camera.Position.y, camera.Position.z};
    for (int k = 0; k < 4; k++) {
      basis[3 * k] = eyeRayBasis[k].x;
      basis[3 * k + 1] = eyeRayBasis[k].y;
      basis[3 * k + 2] = eyeRayBasis[k].z;
    }
    float milliseconds = 0;
    gpuErrchk(pt_mgpu_render(impl, d_buffer.buffer, reinterpret_cast<const pt_sphere*>(d_scene.objects), d_scene.numObjects,
                             basis, eye, &milliseconds));
    return milliseconds;
  }

  std::string Backend() const {
    char name[128] = "";
    gpuErrchk(pt_mgpu_backend(impl, name, sizeof name));
    return name;
  }
  // kernel milliseconds of rank's tile in the last frame
  float TileKernelMs(int rank) const {
    float ms = 0;
    gpuErrchk(pt_mgpu_tile(impl, rank, NULL, NULL, NULL, &ms));
    return ms;
  }
};
#endif
