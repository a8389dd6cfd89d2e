// Denoiser.h -- look-alike of include/Denoiser.h.  The reference's Denoiser only packs the frame
// for its OpenGL point-sprite display (denoise_kernel, src/denoise.cu:9-29); the GLPixelBuffer it
// maps is replaced by a plain device array of width*height vertex triples.
#ifndef DENOISER_H
#define DENOISER_H
#include "HipErrorCheck.h"
#include "OutputBuffer.h"

class Denoiser {
 private:
  int width, height;

 public:
  // Denoiser.h:16-27; numThreads is accepted and ignored like in Renderer
  Denoiser(int width, int height, int numThreads) {
    this->width = width;
    this->height = height;
    (void)numThreads;
  }

  // Denoiser.h:29-52: synchronous like the reference (cudaThreadSynchronize before and after).
  // d_vertices: device float[width*height*3] receiving (col, width - row, RGBA8-in-a-float).
  void Denoise(const OutputBuffer& d_buffer, float* d_vertices) {
    gpuErrchk(pt_device_synchronize());
    gpuErrchk(pt_display_pack(d_buffer.buffer, width, height, d_vertices, NULL));
    gpuErrchk(pt_device_synchronize());
  }
};
#endif
