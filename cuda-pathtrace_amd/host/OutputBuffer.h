// OutputBuffer.h -- look-alike of include/OutputBuffer.h: 14-channel AoS float buffer,
// public width/height/buffer, manual Allocate*/Free*/CopyFromGPU, no destructor, trivially
// copyable (it is passed BY VALUE into Render, Renderer.h:55).
#ifndef OUTPUTBUFFER_H
#define OUTPUTBUFFER_H
#include <string>

#include "ExrWriter.h"
#include "HipErrorCheck.h"

class OutputBuffer {
 public:
  int width, height;
  float* buffer;  // 14 channels, [row][col][14] (pathtrace.cu:240-254)

  OutputBuffer() {  // OutputBuffer.h:36-40
    width = height = -1;
    buffer = NULL;
  }
  OutputBuffer(int width, int height) {  // :42-47
    this->width = width;
    this->height = height;
    buffer = NULL;
  }

  void CopyFromGPU(const OutputBuffer& d_buffer) {  // :49-50
    gpuErrchk(pt_memcpy_d2h(buffer, d_buffer.buffer, (size_t)width * height * 14 * sizeof(float)));
  }
  void AllocateCPU() { buffer = new float[(size_t)width * height * 14]; }  // :61-62
  void AllocateGPU() {                                                       // :73-74
    void* d = NULL;
    gpuErrchk(pt_malloc(&d, (size_t)width * height * 14 * sizeof(float)));
    buffer = static_cast<float*>(d);
  }
  void FreeCPU() { delete[] buffer; }           // :96-97
  void FreeGPU() { (void)pt_free(buffer); }     // :108-109 (unchecked in the reference too)

  // OutputBuffer.h:120-201: 14-channel float32 scanline EXR, uncompressed, channel names and
  // order as the reference writes them (consumed by denoise_cnn/load_data.py:10-18,42-68).
  void SaveEXR(std::string filename) {
    std::string err;
    if (!ptexr::SaveFeatureEXR(filename, buffer, width, height, &err)) {
      fprintf(stderr, "Error saving EXR: %s\n", err.c_str());  // :193
      return;
    }
  }

  // OutputBuffer.h:85-94 + :13-22: eight 8-bit BMPs, (uchar)min(255,max(0,(int)(255*v))).
  void SaveBitmaps(std::string filenameBase) {
    saveFeatureToBitmap(filenameBase + "_color.bmp", 0, 3);
    saveFeatureToBitmap(filenameBase + "_normal.bmp", 3, 3);
    saveFeatureToBitmap(filenameBase + "_albedo.bmp", 6, 3);
    saveFeatureToBitmap(filenameBase + "_depth.bmp", 9, 1);
    saveFeatureToBitmap(filenameBase + "_color_var.bmp", 10, 1);
    saveFeatureToBitmap(filenameBase + "_normal_var.bmp", 11, 1);
    saveFeatureToBitmap(filenameBase + "_albedo_var.bmp", 12, 1);
    saveFeatureToBitmap(filenameBase + "_depth_var.bmp", 13, 1);
  }

 private:
  void saveFeatureToBitmap(std::string filename, int feature, int channels) {
    ptexr::SaveFeatureBMP(filename, buffer, width, height, feature, channels);
  }
};
#endif
