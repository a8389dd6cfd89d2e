"""ctypes binding for the CPU oracle (oracle/libpt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (cuda-pathtrace_amd/) never imports this module.
Parity status: parity unpinned (see oracle/pt_oracle.h).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")

RNG_XORWOW = 0
RNG_PHILOX = 1

SPHERE_DTYPE = np.dtype(
    [("radius", "<f4"), ("pos", "<f4", 3), ("emission", "<f4", 3), ("color", "<f4", 3)]
)
assert SPHERE_DTYPE.itemsize == 40


class Params(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_int32),
        ("height", ctypes.c_int32),
        ("row_begin", ctypes.c_int32),
        ("row_end", ctypes.c_int32),
        ("spp", ctypes.c_int32),
        ("max_bounces", ctypes.c_int32),
        ("rng_mode", ctypes.c_int32),
        ("frame", ctypes.c_uint32),
        ("seed", ctypes.c_uint64),
    ]


def usable_cores():
    """Threads the oracle may really use: the scheduler affinity mask capped by the cgroup CPU quota
    (a GPU box exposes 256 logical CPUs but grants a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "pt_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpt_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_NATIVE_PATH = os.path.join(_HERE, "_native", "libpt_oracle_native.so")
_native = None


def native_lib():
    """The same source built `-O3 -march=native` (gcc's default FP contraction, the host's vector ISA): the second
    CPU-baseline figure of SURVEY.md 8(d).  TIMING ONLY -- it does not follow the numeric contract, so it is never
    compared with anything.  -march=native binds it to the machine that compiled it: built where it runs."""
    global _native
    if _native is None:
        os.makedirs(os.path.dirname(_NATIVE_PATH), exist_ok=True)
        subprocess.check_call(["gcc", "-std=c99", "-O3", "-march=native", "-fPIC", "-pthread", "-shared", "-o", _NATIVE_PATH,
                               os.path.join(_HERE, "pt_oracle.c"), "-lm", "-pthread"])
        _native = _bind(ctypes.CDLL(_NATIVE_PATH))
    return _native


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _bind(ctypes.CDLL(_LIB_PATH))
    return _lib


def _bind(L):
    fp = ctypes.POINTER(ctypes.c_float)
    up = ctypes.POINTER(ctypes.c_uint32)
    L.pto_render.restype = ctypes.c_int
    L.pto_render.argtypes = [ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_int, fp, fp, fp, up, ctypes.c_int]
    L.pto_setup_random.restype = None
    L.pto_setup_random.argtypes = [ctypes.POINTER(Params), up]
    L.pto_scene_cornell.restype = None
    L.pto_scene_cornell.argtypes = [ctypes.c_void_p]
    L.pto_camera_basis.restype = None
    L.pto_camera_basis.argtypes = [fp, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, fp]
    L.pto_camera_basis_up.restype = None
    L.pto_camera_basis_up.argtypes = [fp, ctypes.c_float, ctypes.c_float, fp, ctypes.c_int, ctypes.c_int, fp]
    L.pto_display_pack.restype = None
    L.pto_display_pack.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp]
    L.pto_xorwow_init.restype = None
    L.pto_xorwow_init.argtypes = [ctypes.c_uint64, up]
    L.pto_xorwow_next.restype = ctypes.c_uint32
    L.pto_xorwow_next.argtypes = [up]
    L.pto_uniform_from_u32.restype = ctypes.c_float
    L.pto_uniform_from_u32.argtypes = [ctypes.c_uint32]
    L.pto_philox4x32_10.restype = None
    L.pto_philox4x32_10.argtypes = [up, up, up]
    L.pto_sincos.restype = None
    L.pto_sincos.argtypes = [ctypes.c_float, fp, fp]
    L.pto_intersect_sphere.restype = ctypes.c_int
    L.pto_intersect_sphere.argtypes = [fp, fp, ctypes.c_void_p, fp]
    return L


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _up(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))


def scene_cornell():
    s = np.zeros(9, dtype=SPHERE_DTYPE)
    lib().pto_scene_cornell(s.ctypes.data)
    return s


def camera_basis(pos=(50.0, 52.0, 295.6), yaw=-90.0, pitch=0.0, w=256, h=256, world_up=None):
    p = np.asarray(pos, dtype=np.float32)
    out = np.zeros(12, dtype=np.float32)
    if world_up is None:
        lib().pto_camera_basis(_fp(p), yaw, pitch, w, h, _fp(out))
    else:
        u = np.asarray(world_up, dtype=np.float32)
        lib().pto_camera_basis_up(_fp(p), yaw, pitch, _fp(u), w, h, _fp(out))
    return out


def render(width, height, spp, spheres=None, basis=None, eye=(50.0, 52.0, 295.6), *, row_begin=0,
           row_end=None, max_bounces=5, rng_mode=RNG_XORWOW, seed=0, frame=0, rng_state=None, threads=None, native=False):
    """Render rows [row_begin,row_end) -> float32 array [rows][width][14].  native=True: the -O3 -march=native
    build (timing only, not the contract's arithmetic)."""
    if row_end is None:
        row_end = height
    if spheres is None:
        spheres = scene_cornell()
    if basis is None:
        basis = camera_basis(eye, w=width, h=height)
    if threads is None:
        threads = usable_cores()
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    basis = np.ascontiguousarray(basis, dtype=np.float32).reshape(12)
    eye = np.ascontiguousarray(eye, dtype=np.float32).reshape(3)
    out = np.zeros((row_end - row_begin, width, 14), dtype=np.float32)
    p = Params(width, height, row_begin, row_end, spp, max_bounces, rng_mode, frame, seed)
    st = None
    if rng_state is not None:
        assert rng_state.dtype == np.uint32 and rng_state.size == (row_end - row_begin) * width * 6
        st = _up(rng_state)
    rc = (native_lib() if native else lib()).pto_render(ctypes.byref(p), spheres.ctypes.data, len(spheres), _fp(basis), _fp(eye), _fp(out), st,
                          threads)
    if rc != 0:
        raise ValueError("pto_render: bad arguments")
    return out


def display_pack(img):
    """denoise_kernel (src/denoise.cu:9-29) on a [rows][cols][14] frame -> [rows][cols][3] vertex triples."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape[0], img.shape[1]
    out = np.zeros((h, w, 3), dtype=np.float32)
    lib().pto_display_pack(_fp(img), w, h, _fp(out))
    return out


def setup_random(width, height, row_begin=0, row_end=None, seed=0):
    if row_end is None:
        row_end = height
    st = np.zeros(((row_end - row_begin) * width, 6), dtype=np.uint32)
    p = Params(width, height, row_begin, row_end, 1, 5, RNG_XORWOW, 0, seed)
    lib().pto_setup_random(ctypes.byref(p), _up(st))
    return st


def xorwow_uniforms(seed, n):
    st = np.zeros(6, dtype=np.uint32)
    lib().pto_xorwow_init(seed, _up(st))
    return [lib().pto_uniform_from_u32(lib().pto_xorwow_next(_up(st))) for _ in range(n)]


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().pto_philox4x32_10(_up(c), _up(k), _up(o))
    return o


def sincos(x):
    s = ctypes.c_float()
    c = ctypes.c_float()
    lib().pto_sincos(ctypes.c_float(x), ctypes.byref(s), ctypes.byref(c))
    return s.value, c.value
