/* oracle_san_driver.c -- exercises the CPU checker (oracle/pt_oracle.c) under AddressSanitizer / UBSan:
 * closed and open scenes, both generators, a ragged tile with persisted generator state, an empty scene,
 * one sphere, zero bounces, the camera basis and the display packer.  Exit 0 = ran to the end; the sanitizers
 * abort the process on a finding (-fno-sanitize-recover).  tests/test_sanitizers.py builds and runs it. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pt_oracle.h"

static double checksum(const float* p, size_t n) {
  double s = 0.0;
  for (size_t i = 0; i < n; i++) s += (double)p[i] * (double)((i % 7) + 1);
  return s;
}

int main(void) {
  pto_sphere cornell[9], open_scene[6];
  pto_scene_cornell(cornell);
  const int keep[6] = {0, 2, 4, 6, 7, 8};
  for (int i = 0; i < 6; i++) open_scene[i] = cornell[keep[i]];
  const float eye[3] = {50.0f, 52.0f, 295.6f}, up[3] = {0.0f, 1.0f, 0.0f};
  float basis[12], basis_up[12];
  const int w = 24, h = 17;
  pto_camera_basis(eye, -90.0f, 0.0f, w, h, basis);
  pto_camera_basis_up(eye, -90.0f, 0.0f, up, w, h, basis_up);
  if (memcmp(basis, basis_up, sizeof(basis)) != 0) return 3;
  double total = 0.0;
  for (int mode = 0; mode < 2; mode++) {
    for (int variant = 0; variant < 5; variant++) {
      pto_params p;
      memset(&p, 0, sizeof(p));
      p.width = w;
      p.height = h;
      p.row_begin = variant == 1 ? 3 : 0;
      p.row_end = variant == 1 ? 11 : h;
      p.spp = variant == 4 ? 1 : 3;
      p.max_bounces = variant == 3 ? 0 : 5;
      p.rng_mode = mode;
      p.seed = mode ? 12345u : 0u;
      const pto_sphere* scene = variant == 2 ? open_scene : cornell;
      const int n = variant == 2 ? 6 : (variant == 4 ? 1 : 9);
      const size_t rows = (size_t)(p.row_end - p.row_begin);
      float* out = (float*)malloc(rows * w * 14 * sizeof(float));
      uint32_t* st = (uint32_t*)malloc(rows * w * 6 * sizeof(uint32_t));
      if (!out || !st) return 4;
      pto_setup_random(&p, st);
      for (int frame = 0; frame < 2; frame++) {  /* the second frame continues from the persisted state */
        p.frame = (uint32_t)frame;
        if (pto_render(&p, scene, n, basis, eye, out, mode == 0 ? st : NULL, 2) != 0) return 5;
        total += checksum(out, rows * w * 14);
      }
      if (variant == 0) {
        float* packed = (float*)malloc((size_t)w * h * 3 * sizeof(float));
        if (!packed) return 4;
        pto_display_pack(out, w, h, packed);
        total += checksum(packed, (size_t)w * h * 3);
        free(packed);
      }
      free(out);
      free(st);
    }
  }
  /* an empty scene: every ray escapes */
  {
    pto_params p;
    memset(&p, 0, sizeof(p));
    p.width = p.height = 8;
    p.row_end = 8;
    p.spp = 2;
    p.max_bounces = 5;
    float out[8 * 8 * 14];
    if (pto_render(&p, cornell, 0, basis, eye, out, NULL, 1) != 0) return 6;
    for (int i = 0; i < 8 * 8 * 14; i++)
      if (out[i] != 0.0f) return 7;
  }
  printf("oracle under sanitizers: checksum %.6e\n", total);
  return 0;
}
