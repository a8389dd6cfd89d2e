// Renderer.h -- look-alike of include/Renderer.h: same constructor, Render() and destructor;
// everything behind them is libptcore.so.
#ifndef RENDERER_H
#define RENDERER_H
#include <chrono>
#include <vector>

#include "Camera.h"
#include "HipErrorCheck.h"
#include "pathtrace.h"

class Renderer {
 private:
  int width, height, samplesPerPixel;
  pt_renderer* impl;  // replaces gridSize/dimBlock/d_states/d_eyeRayBasis/d_eyePos (Renderer.h:10-20)

 public:
  // Renderer.h:23-46.  numThreads (threads per block edge, main.cu:21) is accepted and ignored.
  Renderer(int width, int height, int samplesPerPixel, int numThreads) {
    this->width = width;
    this->height = height;
    this->samplesPerPixel = samplesPerPixel;
    impl = NULL;
    gpuErrchk(pt_renderer_create(width, height, samplesPerPixel, numThreads, NULL, &impl));
  }
  // Extension: explicit options (max_bounces, rng_mode, row tile ...).
  Renderer(int width, int height, int samplesPerPixel, int numThreads, const pt_renderer_opts& opts) {
    this->width = width;
    this->height = height;
    this->samplesPerPixel = samplesPerPixel;
    impl = NULL;
    gpuErrchk(pt_renderer_create(width, height, samplesPerPixel, numThreads, &opts, &impl));
  }
  Renderer(const Renderer&) = delete;
  Renderer& operator=(const Renderer&) = delete;

  ~Renderer() { (void)pt_renderer_destroy(impl); }  // Renderer.h:48-53

  // Renderer.h:55-76: synchronous, returns kernel-only milliseconds.
  float Render(OutputBuffer d_buffer, const Scene& d_scene, const Camera& camera) {
    float3 eyeRayBasis[4];
    camera.getEyeRayBasis(eyeRayBasis, width, height);  // :58
    float basis[12], eye[3] = {camera.Position.x, camera.Position.y, camera.Position.z};
    for (int k = 0; k < 4; k++) {
      basis[3 * k] = eyeRayBasis[k].x;
      basis[3 * k + 1] = eyeRayBasis[k].y;
      basis[3 * k + 2] = eyeRayBasis[k].z;
    }
    float milliseconds = 0;
    gpuErrchk(pt_renderer_render(impl, d_buffer.buffer, reinterpret_cast<const pt_sphere*>(d_scene.objects),
                                 d_scene.numObjects, basis, eye, &milliseconds));
    return milliseconds;
  }

  // Extension (no reference counterpart: main.cu:146-148 calls Render once per loop iteration): the frames of a scripted
  // fly-through in one call -- frame k with cameras[k] into d_frames + k * width * height * 14, generator state carried from
  // frame to frame exactly as by Render (pt_renderer_enqueue_frames).  Synchronous; returns wall milliseconds for all frames.
  float RenderFrames(float* d_frames, const Scene& d_scene, const std::vector<Camera>& cameras) {
    std::vector<float> bases(12 * cameras.size()), eyes(3 * cameras.size());
    for (size_t f = 0; f < cameras.size(); f++) {
      float3 eyeRayBasis[4];
      cameras[f].getEyeRayBasis(eyeRayBasis, width, height);
      for (int k = 0; k < 4; k++) {
        bases[12 * f + 3 * k] = eyeRayBasis[k].x;
        bases[12 * f + 3 * k + 1] = eyeRayBasis[k].y;
        bases[12 * f + 3 * k + 2] = eyeRayBasis[k].z;
      }
      eyes[3 * f] = cameras[f].Position.x;
      eyes[3 * f + 1] = cameras[f].Position.y;
      eyes[3 * f + 2] = cameras[f].Position.z;
    }
    gpuErrchk(pt_device_synchronize());
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    gpuErrchk(pt_renderer_enqueue_frames(impl, (int)cameras.size(), d_frames, (size_t)width * height * 14, NULL, 0,
                                         reinterpret_cast<const pt_sphere*>(d_scene.objects), d_scene.numObjects, bases.data(),
                                         eyes.data(), NULL));
    gpuErrchk(pt_device_synchronize());
    gpuErrchk(pt_renderer_check(impl, 1, NULL));
    return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
};
#endif
