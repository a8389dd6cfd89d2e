// pathtrace.h -- look-alike of include/pathtrace.h:10,13.  The reference declares two
// __global__ kernels that only Renderer launches; here the kernels live inside libptcore.so
// (hand-written HIP for gfx950) and the kernel-level entry is the C ABI.
#ifndef PATHTRACE_H
#define PATHTRACE_H
#include "../../include/ptcore.h"
#include "OutputBuffer.h"
#include "Scene.h"
// pixel_kernel(OutputBuffer, curandState*, Scene, float3* rayBasis, float3* eyePos, int spp)
//   -> pt_renderer_render / pt_renderer_enqueue (generator state is owned by pt_renderer,
//      camera travels as kernel arguments)
// setup_random(curandState*, int width, int height)
//   -> runs inside pt_renderer_create / pt_renderer_reset_rng
#endif
