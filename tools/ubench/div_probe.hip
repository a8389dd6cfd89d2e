// probe: the pieces of pt_device.h's div_by_count for a few dividends (debugging aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#pragma clang fp contract(off)
__global__ void probe(const uint32_t* in, uint32_t* out, float nf, int cnt) {
  int i = threadIdx.x;
  if (i >= cnt) return;
  float delta = __uint_as_float(in[i]);
  float y = 1.0f / nf;
  float q0 = delta * y;
  float r = fmaf(-nf, q0, delta);
  float q = fmaf(r, y, q0);
  unsigned cls = (__builtin_amdgcn_class(q0, 0x108) ? 1u : 0u) | (__builtin_amdgcn_class(q0, 0x090) ? 2u : 0u) | (__builtin_amdgcn_class(q0, 0x060) ? 4u : 0u) | (__builtin_isnormal(q0) ? 8u : 0u);
  float slow = delta / nf;
  out[i * 6 + 0] = __float_as_uint(q0);
  out[i * 6 + 1] = __float_as_uint(r);
  out[i * 6 + 2] = __float_as_uint(q);
  out[i * 6 + 3] = cls;
  out[i * 6 + 4] = __float_as_uint(slow);
  out[i * 6 + 5] = __float_as_uint(y);
}
int main() {
  uint32_t h_in[6] = {0x007ff17fu, 0x00fff96fu, 0x807ff12fu, 0x80fff14fu, 0x3f800000u, 0x00000005u};
  uint32_t *d_in, *d_out, h_out[36];
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d_in, d_out, 10.0f, 6);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  for (int i = 0; i < 6; i++) {
    float d; memcpy(&d, &h_in[i], 4);
    float host = d / 10.0f; uint32_t hb; memcpy(&hb, &host, 4);
    printf("delta %08x: q0 %08x r %08x q %08x class %u slow %08x y %08x host %08x\n", h_in[i], h_out[i*6], h_out[i*6+1], h_out[i*6+2], h_out[i*6+3], h_out[i*6+4], h_out[i*6+5], hb);
  }
  return 0;
}
