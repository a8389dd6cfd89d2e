// Test infrastructure (CPU suite): rocRAND's Philox4x32-10 block function (ten_rounds; a third party's implementation, host-compilable)
// on counters and keys read from stdin, for tests/test_generators_rocrand.py to compare with the oracle's pto_philox4x32_10 -- the
// counter-based generator north_star asks for (no reference line: the reference has XORWOW only).
//   stdin: lines "c0 c1 c2 c3 k0 k1"; stdout: "o0 o1 o2 o3" per line
#include <cstdio>
#include <rocrand/rocrand_philox4x32_10.h>

struct Block : rocrand_device::philox4x32_10_engine {
  uint4 run(uint4 c, uint2 k) { return ten_rounds(c, k); }
};

int main() {
  Block b;
  unsigned int c0, c1, c2, c3, k0, k1;
  while (scanf("%u %u %u %u %u %u", &c0, &c1, &c2, &c3, &k0, &k1) == 6) {
    uint4 c; c.x = c0; c.y = c1; c.z = c2; c.w = c3;
    uint2 k; k.x = k0; k.y = k1;
    const uint4 o = b.run(c, k);
    printf("%u %u %u %u\n", o.x, o.y, o.z, o.w);
  }
  return 0;
}
