"""XORWOW against a third party's implementation (CPU suite).

The reference draws from cuRAND's XORWOW (src/pathtrace.cu:131,223-224,265); cuRAND is closed and absent, and the oracle's
restatement (oracle/pt_oracle.c: pto_xorwow_init / pto_xorwow_next) was written by the author of the kernels.  rocRAND ships a
host-compilable engine of the same generator (Marsaglia's xorwow + Weyl sequence) written by someone else: this test injects the
oracle's seeded state into it and compares (i) the 32-bit output stream, (ii) the state after the stream, (iii) rocRAND's
matrix-power skip-ahead `discard(n)` -- computed from precomputed powers of the recurrence matrix, no step loop -- with n oracle
steps, (iv) Marsaglia's base state: the seed that cancels each library's own scramble must give the same stream in both.
What it cannot pin: cuRAND's seed-scramble constants (rocRAND deliberately uses others, rocrand_xorwow.h:113-116) and the
float conversion of curand_uniform -- those stay "recalled" (DESIGN 1.2).
"""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCRAND = "/opt/rocm/include/rocrand/rocrand_xorwow.h"
SRC = os.path.join(ROOT, "tests", "cpp", "rocrand_xorwow_stream.cpp")

pytestmark = pytest.mark.skipif(not os.path.exists(ROCRAND), reason="rocRAND headers not installed")


@pytest.fixture(scope="module")
def engine(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("rocrand") / "rocrand_xorwow_stream")
    res = subprocess.run(["g++", "-O1", "-w", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", exe, SRC],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr

    def run(*args):
        out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, check=True).stdout.split("\n")
        first = [int(x) for x in out[0].split()[1:]]
        last = [int(x) for x in out[-2].split()[1:]]
        return first, [int(x) for x in out[1:-2]], last

    return run


def oracle_state(oracle, seed):
    st = (ctypes.c_uint32 * 6)()
    oracle.lib().pto_xorwow_init(ctypes.c_uint64(seed), st)
    return st


def oracle_stream(oracle, st, n):
    nxt = oracle.lib().pto_xorwow_next
    return [nxt(st) for _ in range(n)]


SEEDS = [0, 1, 255, 65535, 1024 * 1024 - 1, 4096 * 4096 - 1, 0x123456789ABCDEF0, 2**64 - 1]


@pytest.mark.parametrize("seed", SEEDS)
def test_stream_from_the_oracles_seeded_state_equals_rocrands(oracle, engine, seed):
    st = oracle_state(oracle, seed)
    words = list(st)
    first, theirs, last = engine("state", *words, 0, 20000)
    assert first == words
    assert theirs == oracle_stream(oracle, st, 20000)
    assert last == list(st)  # d and the five xorshift words after 20 000 draws


@pytest.mark.parametrize("skip", [1, 2, 5, 72, 1000, 65537, 10**6 + 3])
def test_rocrands_matrix_skip_ahead_equals_that_many_oracle_steps(oracle, engine, skip):
    """discard(n) multiplies the state by precomputed powers A^(4^k) of the recurrence matrix: a different computation of the same
    map, so it would disagree if the oracle's shifts (2 right, 1 and 4 left), its word rotation or the Weyl step were off."""
    st = oracle_state(oracle, 12345)
    words = list(st)
    oracle_stream(oracle, st, skip)
    _, theirs, last = engine("state", *words, skip, 64)
    assert theirs == oracle_stream(oracle, st, 64)
    assert last == list(st)


def test_marsaglias_base_state_is_the_same_in_both_libraries(oracle, engine):
    """Each library xors the seed with its own constants before multiplying: the seed equal to those constants makes t0 = t1 = 0 and
    leaves Marsaglia's x, y, z, w, v = 123456789, 362436069, 521288629, 88675123, 5783321 and d = 6615241 (curand_init's and
    rocRAND's common starting point)."""
    ours = oracle_state(oracle, (0xF7DCEFDD << 32) | 0xAAD26B49)
    first, theirs, _ = engine("seed", 0x2C7F967F, 0xA03697CB, 0, 4096)
    assert list(ours) == first == [6615241, 123456789, 362436069, 521288629, 88675123, 5783321]
    assert theirs == oracle_stream(oracle, ours, 4096)


def test_philox_block_function_equals_rocrands_on_random_counters_and_keys(oracle, tmp_path):
    """Philox4x32-10 (the counter-based generator of north_star; no line of the reference): beyond the published Random123 vectors
    (tests/golden/philox_kats.json) the oracle's block function equals rocRAND's `ten_rounds` on 20 000 random (counter, key) pairs
    and on the all-ones / all-zero / single-bit patterns."""
    import numpy as np
    if not os.path.exists("/opt/rocm/include/rocrand/rocrand_philox4x32_10.h"):
        pytest.skip("rocRAND headers not installed")
    exe = str(tmp_path / "rocrand_philox_block")
    res = subprocess.run(["g++", "-O1", "-w", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", exe,
                          os.path.join(ROOT, "tests", "cpp", "rocrand_philox_block.cpp")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    rs = np.random.RandomState(20261005)
    cases = rs.randint(0, 2**32, size=(20000, 6), dtype=np.uint64).astype(np.uint32)
    special = [[0] * 6, [0xFFFFFFFF] * 6] + [[(1 << b) if w == j else 0 for j in range(6)] for w in range(6) for b in (0, 15, 31)]
    cases = np.concatenate([np.array(special, dtype=np.uint32), cases])
    text = "\n".join(" ".join(str(int(v)) for v in row) for row in cases) + "\n"
    out = subprocess.run([exe], input=text, capture_output=True, text=True, check=True).stdout
    theirs = np.array([[int(v) for v in line.split()] for line in out.strip().split("\n")], dtype=np.uint32)
    assert theirs.shape == (len(cases), 4)
    ours = np.stack([oracle.philox(row[:4], row[4:6]) for row in cases])
    assert np.array_equal(ours, theirs)
