/* abi_example.c -- the "raw C binding" of INTEGRATION.md as a strict C99 program: include/ptcore.h must be
 * plain C, and without a GPU the very first call fails loudly (exit 3) instead of falling back.
 * With an argument the rendered frame is written to that file as raw float32 [64][64][14]. */
#include <stdio.h>
#include <stdlib.h>
#include "include/ptcore.h"
int main(int argc, char** argv) {
  pt_renderer* r; float ms, basis[12], eye[3] = {50.f, 52.f, 295.6f};
  pt_sphere host[9], *d_spheres; float* d_out; int W = 64, H = 64, spp = 4;
  size_t bytes = (size_t)W * H * 14 * sizeof(float);
  if (pt_set_device(0) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 3; }
  pt_scene_cornell(host);
  pt_malloc((void**)&d_spheres, sizeof host); pt_memcpy_h2d(d_spheres, host, sizeof host);
  pt_malloc((void**)&d_out, bytes);
  pt_camera_basis(eye, -90.f, 0.f, W, H, basis);
  pt_renderer_create(W, H, spp, 8, NULL, &r);
  if (pt_renderer_render(r, d_out, d_spheres, 9, basis, eye, &ms) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }
  printf("ok %f ms abi %d\n", ms, pt_abi_version());
  if (argc > 1) {
    float* h = (float*)malloc(bytes); FILE* f = fopen(argv[1], "wb");
    if (!h || !f || pt_memcpy_d2h(h, d_out, bytes) != PT_OK || fwrite(h, 1, bytes, f) != bytes) return 2;
    fclose(f); free(h);
  }
  pt_renderer_destroy(r); pt_free(d_out); pt_free(d_spheres);
  return 0;
}
