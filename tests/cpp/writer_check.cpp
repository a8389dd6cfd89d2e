// writer_check.cpp -- writes the same synthetic buffers as oracle/ref_writers/ref_writers.cpp
// through the product's OutputBuffer look-alike writers (host-only code path: AllocateCPU /
// SaveEXR / SaveBitmaps need no GPU).  The test compares the files byte for byte with the
// fixtures produced by the reference's own tinyexr / stb writers.
#include <math.h>
#include <stdlib.h>

#include <string>

#include "ExrWriter.h"

static float pattern(int i, int mode) {
  if (mode == 0) return (float)i;
  return 1.3f * sinf(0.37f * (float)i) + 0.002f * (float)(i % 97);
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int width = atoi(argv[1]), height = atoi(argv[2]), mode = atoi(argv[3]);
  const std::string out = argv[4];
  std::vector<float> buffer((size_t)width * height * 14);
  for (size_t i = 0; i < buffer.size(); i++) buffer[i] = pattern((int)i, mode);
  std::string err;
  if (!ptexr::SaveFeatureEXR(out + ".exr", buffer.data(), width, height, &err)) return 1;
  if (!ptexr::SaveFeatureBMP(out + "_color.bmp", buffer.data(), width, height, 0, 3)) return 1;
  if (!ptexr::SaveFeatureBMP(out + "_depth.bmp", buffer.data(), width, height, 9, 1)) return 1;
  return 0;
}
