// ref_writers.cpp -- drives the reference's OWN vendored writers (include/tinyexr.h and
// include/stb_image_write.h, compiled in place from /root/reference, nothing copied) with the
// arguments OutputBuffer::SaveEXR / saveFeatureToBitmap pass (include/OutputBuffer.h:13-22,
// 120-201), to produce byte-exact fixtures for the EXR/BMP writer of cuda-pathtrace_amd/host.
// TEST INFRASTRUCTURE: built only when /root/reference exists, output under oracle/_ref/.
// Usage: ref_writers <width> <height> <out_prefix>   (buffer[i] = pattern(i), see fill()).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#define TINYEXR_IMPLEMENTATION
#include "tinyexr.h"
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "stb_image_write.h"

static float pattern(int i, int mode) {
  if (mode == 0) return (float)i;                                   // SURVEY's 8x8 ramp
  return 1.3f * sinf(0.37f * (float)i) + 0.002f * (float)(i % 97);  // exercises clamping / negatives
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int width = atoi(argv[1]), height = atoi(argv[2]), mode = atoi(argv[3]);
  const std::string out = argv[4];
  std::vector<float> buffer((size_t)width * height * 14);
  for (size_t i = 0; i < buffer.size(); i++) buffer[i] = pattern((int)i, mode);

  // ---- OutputBuffer::SaveEXR, OutputBuffer.h:120-201 (same calls, same argument values) ----
  EXRHeader header;
  InitEXRHeader(&header);
  EXRImage image;
  InitEXRImage(&image);
  image.num_channels = 14;
  std::vector<float> images[14];
  for (int c = 0; c < 14; c++) images[c].resize((size_t)width * height);
  int i = 0;
  for (int x = 0; x < width; x++)
    for (int y = 0; y < height; y++) {
      for (int c = 0; c < 14; c++) images[c][i] = buffer[x * width * 14 + y * 14 + c];
      i++;
    }
  const int src[14] = {8, 7, 6, 12, 2, 1, 0, 10, 9, 13, 5, 4, 3, 11};
  const char* names[14] = {"Albedo.B", "Albedo.G", "Albedo.R", "AlbedoVar.Z", "Color.B", "Color.G", "Color.R",
                           "ColorVar.Z", "Depth.Z", "DepthVar.Z", "Normal.Z", "Normal.Y", "Normal.X", "NormalVar.Z"};
  float* image_ptr[14];
  for (int c = 0; c < 14; c++) image_ptr[c] = &images[src[c]].at(0);
  image.images = (unsigned char**)image_ptr;
  image.width = width;
  image.height = height;
  header.num_channels = 14;
  header.channels = (EXRChannelInfo*)malloc(sizeof(EXRChannelInfo) * 14);
  for (int c = 0; c < 14; c++) {
    strncpy(header.channels[c].name, names[c], 255);
    header.channels[c].name[strlen(names[c])] = '\0';
  }
  header.pixel_types = (int*)malloc(sizeof(int) * 14);
  header.requested_pixel_types = (int*)malloc(sizeof(int) * 14);
  for (int c = 0; c < 14; c++) header.pixel_types[c] = header.requested_pixel_types[c] = TINYEXR_PIXELTYPE_FLOAT;
  const char* err = NULL;
  if (SaveEXRImageToFile(&image, &header, (out + ".exr").c_str(), &err) != TINYEXR_SUCCESS) {
    fprintf(stderr, "Error saving EXR: %s\n", err);
    return 1;
  }

  // ---- saveFeatureToBitmap, OutputBuffer.h:13-22, for the 3-channel colour and 1-channel depth ----
  const int feats[2][2] = {{0, 3}, {9, 1}};
  const char* suffix[2] = {"_color.bmp", "_depth.bmp"};
  for (int k = 0; k < 2; k++) {
    const int feature = feats[k][0], channels = feats[k][1];
    std::vector<unsigned char> ob((size_t)width * height * channels);
    for (int x = 0; x < width; x++)
      for (int y = 0; y < height; y++)
        for (int c = 0; c < channels; c++)
          ob[x * width * channels + y * channels + c] =
              (unsigned char)std::min(255, std::max(0, (int)(255.0f * buffer[x * width * 14 + y * 14 + feature + c])));
    stbi_write_bmp((out + suffix[k]).c_str(), width, height, channels, ob.data());
  }
  return 0;
}
