// Test infrastructure (CPU suite): drives rocRAND's XORWOW engine -- a third party's implementation of Marsaglia's xorwow, the
// generator cuRAND calls XORWOW (src/pathtrace.cu:131,223-224,265 draw from it) -- from a state given on the command line, so
// that tests/test_generators_rocrand.py can compare its stream and its matrix-power skip-ahead with the oracle's restatement.
// rocRAND's header is host-compilable (plain g++); its SEED SCRAMBLE uses other constants than cuRAND's
// (/opt/rocm/include/rocrand/rocrand_xorwow.h:113-116), so only the recurrence, the Weyl step and Marsaglia's base state can be
// cross-checked here, not curand_init's scramble.
//   usage: rocrand_xorwow_stream state d x0 x1 x2 x3 x4 skip count     -> `count` outputs after discarding `skip`
//          rocrand_xorwow_stream seed  lo hi skip count                -> the same from rocRAND's own seeding of (hi << 32 | lo)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <rocrand/rocrand_xorwow.h>

struct Engine : rocrand_device::xorwow_engine {
  Engine(unsigned long long seed) : rocrand_device::xorwow_engine(seed, 0, 0) {}
  void set(const unsigned int* st) {
    m_state.d = st[0];
    for (int i = 0; i < 5; i++) m_state.x[i] = st[1 + i];
  }
  void show() const { printf("state %u %u %u %u %u %u\n", m_state.d, m_state.x[0], m_state.x[1], m_state.x[2], m_state.x[3], m_state.x[4]); }
};

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  unsigned long long skip = 0, count = 0;
  Engine e(0);
  if (!strcmp(argv[1], "state") && argc == 10) {
    unsigned int st[6];
    for (int i = 0; i < 6; i++) st[i] = (unsigned int)strtoul(argv[2 + i], 0, 0);
    e.set(st);
    skip = strtoull(argv[8], 0, 0);
    count = strtoull(argv[9], 0, 0);
  } else if (!strcmp(argv[1], "seed") && argc == 6) {
    e = Engine((strtoull(argv[3], 0, 0) << 32) | strtoull(argv[2], 0, 0));
    skip = strtoull(argv[4], 0, 0);
    count = strtoull(argv[5], 0, 0);
  } else {
    return 2;
  }
  e.show();
  if (skip) e.discard(skip);  // jump matrices A^(4^k): not the step loop
  for (unsigned long long i = 0; i < count; i++) printf("%u\n", e.next());
  e.show();
  return 0;
}
