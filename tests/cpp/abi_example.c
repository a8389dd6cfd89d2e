/* abi_example.c -- the "raw C binding" of INTEGRATION.md as a strict C99 program: include/ptcore.h must be
 * plain C, and without a GPU the very first call fails loudly (exit 3) instead of falling back. */
#include <stdio.h>
#include "include/ptcore.h"
int main(void) {
  pt_renderer* r; float ms, basis[12], eye[3] = {50.f, 52.f, 295.6f};
  pt_sphere host[9], *d_spheres; float* d_out; int W = 64, H = 64, spp = 4;
  if (pt_set_device(0) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 3; }
  pt_scene_cornell(host);
  pt_malloc((void**)&d_spheres, sizeof host); pt_memcpy_h2d(d_spheres, host, sizeof host);
  pt_malloc((void**)&d_out, (size_t)W * H * 14 * sizeof(float));
  pt_camera_basis(eye, -90.f, 0.f, W, H, basis);
  pt_renderer_create(W, H, spp, 8, NULL, &r);
  if (pt_renderer_render(r, d_out, d_spheres, 9, basis, eye, &ms) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }
  printf("ok %f ms abi %d\n", ms, pt_abi_version());
  pt_renderer_destroy(r); pt_free(d_out); pt_free(d_spheres);
  return 0;
}
