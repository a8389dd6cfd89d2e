"""cuda-pathtrace_amd -- MI355X (gfx950) implementation of cuda-pathtrace's per-pixel
Monte-Carlo megakernel, behind the C ABI of include/ptcore.h.

This Python module is only the ctypes view of libptcore.so used by tests/, bench.py and the
multi-GPU driver (one process per GPU under torch.distributed).  The product is the shared
library and the C++ look-alike headers in host/; nothing here computes pixels, and there is
no CPU fallback: importing works without a GPU (the library loads), every compute call
raises PtError when no HIP device is usable, and a missing libptcore.so raises at import.

The directory name has a hyphen (it is fixed by the project layout), so import it through
`load_package()` in __graft_entry__.py / tests/conftest.py, which registers it as
`cuda_pathtrace_amd`.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_LIB_OVERRIDE: another build of the same library -- A/B experiments (tools/ab_tiles.py) and the lab build
# libptcore_lab.so (experimental kernel variants + pt_debug_* diagnostics; __graft_entry__.load_lab())
LIB_PATH = os.environ.get("PT_LIB_OVERRIDE") or os.path.join(_HERE, "libptcore.so")
LAB_LIB_PATH = os.path.join(_HERE, "libptcore_lab.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C cuda-pathtrace_amd/csrc` "
        "(or __graft_entry__.build()); there is no fallback implementation"
    )

RNG_XORWOW = 0
RNG_PHILOX = 1
CHANNELS = 14

SPHERE_DTYPE = np.dtype([("radius", "<f4"), ("pos", "<f4", 3), ("emission", "<f4", 3), ("color", "<f4", 3)])
assert SPHERE_DTYPE.itemsize == 40


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ptcore error {code}: {msg}")
        self.code = code


class RendererOpts(ctypes.Structure):
    _fields_ = [
        ("max_bounces", ctypes.c_int32),
        ("rng_mode", ctypes.c_int32),
        ("seed", ctypes.c_uint64),
        ("row_begin", ctypes.c_int32),
        ("row_end", ctypes.c_int32),
        ("persist_rng", ctypes.c_int32),
        ("variant", ctypes.c_int32),
        ("layout", ctypes.c_int32),
        ("fast_math", ctypes.c_int32),
        ("chunks", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


VARIANT_FAST = 100
LAYOUT_INTERLEAVED, LAYOUT_PLANAR = 0, 1
GATHER_AUTO, GATHER_RCCL, GATHER_PEER_COPY = 0, 1, 2


class MgpuOpts(ctypes.Structure):
    _fields_ = [
        ("gather", ctypes.c_int32),
        ("force_exchange", ctypes.c_int32),
        ("timeout_ms", ctypes.c_int32),
        ("bands", ctypes.c_int32),
    ]


class KernelInfo(ctypes.Structure):
    _fields_ = [
        ("block_threads", ctypes.c_int32),
        ("grid_blocks", ctypes.c_int32),
        ("lds_bytes", ctypes.c_int32),
        ("num_vgprs", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
        ("scratch_bytes", ctypes.c_int32),
        ("max_spheres", ctypes.c_int32),
        ("variant", ctypes.c_int32),
    ]


_fp = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p

# name -> (restype, argtypes); this table is also what tests check against include/ptcore.h
ABI = {
    "pt_abi_version": (ctypes.c_int, []),
    "pt_build_fingerprint": (ctypes.c_char_p, []),
    "pt_last_error": (ctypes.c_char_p, []),
    "pt_set_device": (ctypes.c_int, [ctypes.c_int]),
    "pt_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "pt_device_info": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "pt_malloc": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_size_t]),
    "pt_free": (ctypes.c_int, [_vp]),
    "pt_memcpy_h2d": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "pt_memcpy_d2h": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "pt_memset": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_size_t]),
    "pt_device_synchronize": (ctypes.c_int, []),
    "pt_renderer_opts_default": (None, [ctypes.POINTER(RendererOpts)]),
    "pt_renderer_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(RendererOpts), ctypes.POINTER(_vp)]),
    "pt_renderer_destroy": (ctypes.c_int, [_vp]),
    "pt_renderer_render": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _fp, _fp, _fp]),
    "pt_renderer_enqueue": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _fp, _fp, _vp]),
    "pt_renderer_enqueue_frames": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp, ctypes.c_int, _fp, _fp, _vp]),
    "pt_renderer_check": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]),
    "pt_renderer_set_display": (ctypes.c_int, [_vp, _vp]),
    "pt_renderer_set_frame": (ctypes.c_int, [_vp, ctypes.c_uint32]),
    "pt_renderer_reset_rng": (ctypes.c_int, [_vp]),
    "pt_renderer_get_rng_state": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "pt_renderer_set_rng_state": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "pt_renderer_kernel_info": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(KernelInfo)]),
    "pt_mgpu_opts_default": (None, [ctypes.POINTER(MgpuOpts)]),
    "pt_mgpu_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.POINTER(RendererOpts), ctypes.POINTER(MgpuOpts), ctypes.POINTER(_vp)]),
    "pt_mgpu_destroy": (ctypes.c_int, [_vp]),
    "pt_mgpu_render": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _fp, _fp, _fp]),
    "pt_mgpu_tile": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                    ctypes.POINTER(ctypes.c_int), _fp]),
    "pt_mgpu_backend": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t]),
    "pt_mgpu_frame_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), _fp, _fp]),
    "pt_scene_cornell": (ctypes.c_int, [_vp]),
    "pt_scene_random": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, _vp]),
    "pt_camera_basis": (ctypes.c_int, [_fp, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, _fp]),
    "pt_camera_basis_up": (ctypes.c_int, [_fp, ctypes.c_float, ctypes.c_float, _fp, ctypes.c_int, ctypes.c_int, _fp]),
    "pt_display_pack": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp]),
}
# include/ptcore_lab.h: only libptcore_lab.so exports these
LAB_ABI = {
    "pt_debug_unary_map": (ctypes.c_int, [ctypes.c_int, _vp, _vp, ctypes.c_size_t]),
    "pt_debug_unary_compare": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64,
                                              ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]),
    "pt_debug_div_compare": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64,
                                            ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32),
                                            ctypes.POINTER(ctypes.c_uint32)]),
    "pt_debug_grid_header": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]),
    "pt_debug_grid_image": (ctypes.c_int, [_vp, ctypes.c_int, _fp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t,
                                           ctypes.POINTER(ctypes.c_uint64)]),
    "pt_debug_policy_ms": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "pt_debug_policy_choice": (ctypes.c_int, [ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
}

FN_INV_SQRT_LITERAL, FN_INV_SQRT_FAST, FN_SQRT_LITERAL, FN_SQRT_FAST, FN_SIN, FN_COS, FN_UNIFORM = range(7)
FN_ONEMINUS_LITERAL, FN_ONEMINUS_FAST, FN_ONEMINUS_F32, FN_ONEMINUS_F32_FLAG, FN_ZERO = 7, 8, 9, 10, 11
FN_UNIFORM_LITERAL = 12
FN_SIN_LITERAL, FN_COS_LITERAL = 13, 14

lib = ctypes.CDLL(LIB_PATH)
for _name, (_res, _args) in ABI.items():
    if os.environ.get("PT_LIB_OVERRIDE") and not hasattr(lib, _name):
        continue  # A/B tooling only: an alternative build of an older ABI (tools/build_alt.sh from an older tree)
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export the ABI
    _fn.restype = _res
    _fn.argtypes = _args
IS_LAB = hasattr(lib, "pt_debug_unary_map")
if IS_LAB:
    for _name, (_res, _args) in LAB_ABI.items():
        _fn = getattr(lib, _name)
        _fn.restype = _res
        _fn.argtypes = _args


def variants():
    """Kernel variants compiled into the loaded library (product: 0, 6, 8, 9, 10, 13, 14; lab: 0..14)."""
    out = []
    o = RendererOpts()
    for v in range(15):
        lib.pt_renderer_opts_default(ctypes.byref(o))
        o.variant = v
        h = _vp()
        rc = lib.pt_renderer_create(8, 8, 1, 8, ctypes.byref(o), ctypes.byref(h))
        if rc == 0:
            lib.pt_renderer_destroy(h)
            out.append(v)
        elif rc != -1:  # anything but "not in this build" is a real failure (no device ...)
            check(rc)
    return out


def build_fingerprint():
    return lib.pt_build_fingerprint().decode()


def check(rc):
    if rc != 0:
        raise PtError(rc, lib.pt_last_error().decode("utf-8", "replace"))


def _f32(a, n):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(n)
    return a, a.ctypes.data_as(_fp)


# ---- host-side inputs -----------------------------------------------------------------
def scene_cornell():
    s = np.zeros(9, dtype=SPHERE_DTYPE)
    check(lib.pt_scene_cornell(s.ctypes.data))
    return s


def scene_random(n, seed=0, with_walls=True):
    s = np.zeros(n, dtype=SPHERE_DTYPE)
    check(lib.pt_scene_random(n, seed, 1 if with_walls else 0, s.ctypes.data))
    return s


DEFAULT_EYE = (50.0, 52.0, 295.6)  # src/main.cu:24


def camera_basis(pos=DEFAULT_EYE, yaw=-90.0, pitch=0.0, width=512, height=512, world_up=None):
    p, pp = _f32(pos, 3)
    out = np.zeros(12, dtype=np.float32)
    if world_up is None:
        check(lib.pt_camera_basis(pp, yaw, pitch, width, height, out.ctypes.data_as(_fp)))
    else:
        u, up = _f32(world_up, 3)
        check(lib.pt_camera_basis_up(pp, yaw, pitch, up, width, height, out.ctypes.data_as(_fp)))
    return out


# ---- device ---------------------------------------------------------------------------
def device_count():
    n = ctypes.c_int(0)
    check(lib.pt_device_count(ctypes.byref(n)))
    return n.value


def set_device(i):
    check(lib.pt_set_device(i))


def device_info():
    name = ctypes.create_string_buffer(256)
    cus = ctypes.c_int(0)
    khz = ctypes.c_int(0)
    check(lib.pt_device_info(name, 256, ctypes.byref(cus), ctypes.byref(khz)))
    return {"name": name.value.decode(), "compute_units": cus.value, "clock_khz": khz.value}


def display_pack(frame):
    """pt_display_pack on a host [rows][cols][14] frame (uploads, packs on the GPU, downloads)."""
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    h, w = frame.shape[0], frame.shape[1]
    din, dout = DeviceBuffer(frame.nbytes).upload(frame), DeviceBuffer(h * w * 12)
    try:
        check(lib.pt_display_pack(din.ptr, w, h, dout.ptr, None))
        check(lib.pt_device_synchronize())
        return dout.download(np.float32, (h, w, 3))
    finally:
        din.free()
        dout.free()


def unary_map(fn, x):
    """Evaluate device building block `fn` on a float32 array (on the GPU)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    din, dout = DeviceBuffer(max(x.nbytes, 4)).upload(x), DeviceBuffer(max(x.nbytes, 4))
    try:
        check(lib.pt_debug_unary_map(fn, din.ptr, dout.ptr, x.size))
        return dout.download(np.float32, x.shape)
    finally:
        din.free()
        dout.free()


def unary_compare(fn_a, fn_b, first_bits=0, count=1 << 32):
    n, ex = ctypes.c_uint64(0), ctypes.c_uint32(0)
    check(lib.pt_debug_unary_compare(fn_a, fn_b, first_bits, count, ctypes.byref(n), ctypes.byref(ex)))
    return n.value, ex.value

def div_compare(n_first, n_count, first_bits=0, count=1 << 32):
    """Lab library: the kernels' division by a sample count (table reciprocal + exact remainder + one correction) against the
    division itself for counts n_first .. n_first + n_count - 1 and `count` dividend bit patterns from first_bits.
    Returns (mismatches, example dividend bits, example count)."""
    n, ex, exn = ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_uint32()
    check(lib.pt_debug_div_compare(n_first, n_count, first_bits, count, ctypes.byref(n), ctypes.byref(ex), ctypes.byref(exn)))
    return n.value, ex.value, exn.value


class DeviceBuffer:
    """hipMalloc'd bytes (OutputBuffer::AllocateGPU / Scene's sphere upload)."""

    def __init__(self, nbytes):
        p = _vp()
        check(lib.pt_malloc(ctypes.byref(p), nbytes))
        self.ptr = p.value
        self.nbytes = nbytes

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib.pt_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(lib.pt_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            check(lib.pt_free(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def policy_ms(rng_mode, variant, waves_per_simd, spp, bounces=5):
    """Lab library: the automatic policy's predicted kernel ms for variant 6 / 8 / 9 on a tile (pt_debug_policy_ms)."""
    ms = ctypes.c_double(0)
    check(lib.pt_debug_policy_ms(rng_mode, variant, waves_per_simd, spp, bounces, ctypes.byref(ms)))
    return ms.value


def policy_choice(rng_mode, waves_per_simd, spp, bounces=5, with9=True):
    """... and the variant the library's OWN policy code (cheapest_variant, csrc/pt_capi.hip) picks from it for the reference's
    scene, in the regime a default renderer is in (samples chunked from 512 spp on): pt_debug_policy_choice."""
    v = ctypes.c_int(0)
    check(lib.pt_debug_policy_choice(rng_mode, waves_per_simd, spp, bounces, 1 if with9 else 0, 1 if spp >= 512 else 0, ctypes.byref(v)))
    return v.value


def grid_header(spheres):
    """Diagnostics: the header of kernel variant 11's uniform grid for a scene (pt_debug_grid_header)."""
    d_scene, n = upload_scene(spheres)
    raw = (ctypes.c_uint32 * 16)()
    check(lib.pt_debug_grid_header(d_scene.ptr, n, raw))
    u = np.frombuffer(raw, dtype=np.uint32).copy()
    f = u.view(np.float32)
    return {"valid": int(u[0]), "dims": (int(u[1]), int(u[2]), int(u[3])), "origin": (float(f[4]), float(f[5]), float(f[6])),
            "cell_size": float(f[7]), "slack": float(f[9]), "centre": (float(f[10]), float(f[11]), float(f[12])),
            "far2": float(f[13]), "n_big": int(u[14]) & 0xFFFF, "n_entries": int(u[14]) >> 16, "n_items": int(u[15])}


def grid_image(spheres, eye=None, threads=512):
    """Diagnostics: everything the grid builder writes for a scene (pt_debug_grid_image), decoded: the header fields, the spheres
    outside the grid, per cell the registered spheres (from the cell starts + registration list) and per cell the spheres the
    pooled walk's table lists (inline entries and chained ones followed)."""
    d_scene, n = upload_scene(spheres)
    lay = (ctypes.c_uint64 * 8)()
    check(lib.pt_debug_grid_image(d_scene.ptr, n, None, threads, None, 0, lay))
    raw = (ctypes.c_uint32 * (int(lay[0]) // 4))()
    eye_c = (ctypes.c_float * 3)(*eye) if eye is not None else None
    check(lib.pt_debug_grid_image(d_scene.ptr, n, eye_c, threads, raw, int(lay[0]), lay))
    u = np.frombuffer(raw, dtype=np.uint32).copy()
    f, b = u.view(np.float32), u.view(np.uint8)
    hdr = {"valid": int(u[0]), "dims": (int(u[1]), int(u[2]), int(u[3])), "origin": (float(f[4]), float(f[5]), float(f[6])),
           "cell_size": float(f[7]), "slack": float(f[9]), "centre": (float(f[10]), float(f[11]), float(f[12])), "far2": float(f[13]),
           "n_big": int(u[14]) & 0xFFFF, "n_entries": int(u[14]) >> 16, "n_items": int(u[15]), "r_small": float(f[16]),
           "r_big": float(f[17]), "max_entries": int(lay[7])}
    if not hdr["valid"]:
        return hdr
    ncells = hdr["dims"][0] * hdr["dims"][1] * hdr["dims"][2]
    u16 = lambda off, cnt: b[off:off + 2 * cnt].view(np.uint16).astype(np.int64)
    hdr["big"] = u16(int(lay[1]), hdr["n_big"])
    start = u16(int(lay[2]), ncells + 1)
    items = u16(int(lay[3]), hdr["n_items"])
    hdr["cells"] = [items[start[c]:start[c + 1]] for c in range(ncells)]
    tab = b[int(lay[4]):int(lay[4]) + 8 * hdr["n_entries"]].view(np.uint32).reshape(-1, 2)
    pooled = []
    for c in range(ncells):
        got, e = [], c
        for _ in range(hdr["n_entries"] + 1):
            w0, w1 = int(tab[e, 0]), int(tab[e, 1])
            k, link = w0 >> 30, (w0 >> 16) & 0x1FFF
            got += [w0 & 0xFFFF, w1 & 0xFFFF, w1 >> 16][:k]
            if link == 0:
                break
            e = link
        pooled.append(np.asarray(got, dtype=np.int64))
    hdr["pooled"] = pooled
    return hdr


def upload_scene(spheres):
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    return DeviceBuffer(max(spheres.nbytes, 4)).upload(spheres), len(spheres)


class Renderer:
    """ctypes view of pt_renderer (the reference's class Renderer, include/Renderer.h)."""

    def __init__(self, width, height, spp, threads_per_block=8, *, max_bounces=5, rng_mode=RNG_XORWOW, seed=0,
                 row_begin=0, row_end=0, persist_rng=True, variant=None, layout=LAYOUT_INTERLEAVED, fast_math=False, chunks=0):
        o = RendererOpts()
        lib.pt_renderer_opts_default(ctypes.byref(o))
        o.max_bounces, o.rng_mode, o.seed = max_bounces, rng_mode, seed
        o.row_begin, o.row_end = row_begin, row_end
        o.persist_rng = 1 if persist_rng else 0
        o.layout = layout
        o.fast_math = 1 if fast_math else 0
        o.chunks = chunks  # 0 automatic, 1 off, n >= 2 chunks
        if variant is not None:
            o.variant = variant
        self.variant = o.variant
        h = _vp()
        check(lib.pt_renderer_create(width, height, spp, threads_per_block, ctypes.byref(o), ctypes.byref(h)))
        self.handle = h.value
        self.width, self.height, self.spp = width, height, spp
        self.row_begin = row_begin
        self.row_end = row_end if (row_begin or row_end) else height
        self.rows = self.row_end - self.row_begin

    @property
    def tile_floats(self):
        return self.rows * self.width * CHANNELS

    def render(self, d_out, d_spheres, n_spheres, basis, eye=DEFAULT_EYE):
        """Synchronous Render(); returns kernel milliseconds (Renderer.h:55-76)."""
        _, b = _f32(basis, 12)
        _, e = _f32(eye, 3)
        ms = ctypes.c_float(0)
        check(lib.pt_renderer_render(self.handle, d_out, d_spheres, n_spheres, b, e, ctypes.byref(ms)))
        return ms.value

    def enqueue(self, d_out, d_spheres, n_spheres, basis, eye=DEFAULT_EYE, stream=None):
        _, b = _f32(basis, 12)
        _, e = _f32(eye, 3)
        check(lib.pt_renderer_enqueue(self.handle, d_out, d_spheres, n_spheres, b, e, stream))

    def enqueue_frames(self, d_out, out_stride_floats, d_spheres, n_spheres, bases, eyes, d_vertices=None, vtx_stride_floats=0, stream=None):
        """n frames of known cameras (bases [n][12], eyes [n][3]) into d_out + f * out_stride_floats: pt_renderer_enqueue_frames."""
        bases = np.ascontiguousarray(bases, dtype=np.float32).reshape(-1, 12)
        eyes = np.ascontiguousarray(eyes, dtype=np.float32).reshape(-1, 3)
        assert len(bases) == len(eyes)
        check(lib.pt_renderer_enqueue_frames(self.handle, len(bases), d_out, out_stride_floats, d_vertices, vtx_stride_floats, d_spheres,
                                             n_spheres, bases.ctypes.data_as(_fp), eyes.ctypes.data_as(_fp), stream))

    def check(self, wait=True):
        """Status of the frames enqueued so far (raises PtError(PT_EKERNEL) for a frame whose sample-chunk chain broke);
        returns the number of frames render() has repaired in place."""
        n = ctypes.c_uint32(0)
        check(lib.pt_renderer_check(self.handle, 1 if wait else 0, ctypes.byref(n)))
        return n.value

    def set_display(self, d_vertices):
        """Every frame also writes the display vertices (Denoiser::Denoise fused into the render); None switches it off."""
        check(lib.pt_renderer_set_display(self.handle, d_vertices))

    def set_frame(self, frame):
        check(lib.pt_renderer_set_frame(self.handle, frame))

    def reset_rng(self):
        check(lib.pt_renderer_reset_rng(self.handle))

    def get_rng_state(self):
        st = np.zeros((self.rows * self.width, 6), dtype=np.uint32)
        check(lib.pt_renderer_get_rng_state(self.handle, st.ctypes.data, st.size))
        return st

    def set_rng_state(self, st):
        st = np.ascontiguousarray(st, dtype=np.uint32)
        check(lib.pt_renderer_set_rng_state(self.handle, st.ctypes.data, st.size))

    def kernel_info(self, n_spheres):
        ki = KernelInfo()
        check(lib.pt_renderer_kernel_info(self.handle, n_spheres, ctypes.byref(ki)))
        return {f: getattr(ki, f) for f, _ in KernelInfo._fields_ if not f.startswith("reserved")}

    def destroy(self):
        if self.handle:
            check(lib.pt_renderer_destroy(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class MultiRenderer:
    """ctypes view of pt_mgpu: one frame row-tiled over several devices of this process (one host thread per
    device inside the library, RCCL or peer-copy exchange to devices[0])."""

    def __init__(self, devices, width, height, spp, threads_per_block=8, *, max_bounces=5, rng_mode=RNG_XORWOW, seed=0,
                 persist_rng=True, variant=None, gather=None, force_exchange=None, timeout_ms=None, fast_math=False, bands=None):
        o = RendererOpts()
        lib.pt_renderer_opts_default(ctypes.byref(o))
        o.max_bounces, o.rng_mode, o.seed = max_bounces, rng_mode, seed
        o.persist_rng = 1 if persist_rng else 0
        o.fast_math = 1 if fast_math else 0
        if variant is not None:
            o.variant = variant
        mo = MgpuOpts()
        lib.pt_mgpu_opts_default(ctypes.byref(mo))
        if gather is not None:
            mo.gather = gather
        if force_exchange is not None:
            mo.force_exchange = 1 if force_exchange else 0
        if timeout_ms is not None:
            mo.timeout_ms = timeout_ms
        if bands is not None:
            mo.bands = bands  # 0 automatic, n row bands per tile (pipelined exchange)
        devs = (ctypes.c_int * len(devices))(*devices)
        h = _vp()
        check(lib.pt_mgpu_create(len(devices), devs, width, height, spp, threads_per_block, ctypes.byref(o), ctypes.byref(mo), ctypes.byref(h)))
        self.handle = h.value
        self.n, self.width, self.height, self.spp = len(devices), width, height, spp

    def render(self, d_out, d_spheres, n_spheres, basis, eye=DEFAULT_EYE):
        """Synchronous; returns end-to-end wall milliseconds (render + exchange)."""
        _, b = _f32(basis, 12)
        _, e = _f32(eye, 3)
        ms = ctypes.c_float(0)
        check(lib.pt_mgpu_render(self.handle, d_out, d_spheres, n_spheres, b, e, ctypes.byref(ms)))
        return ms.value

    def tile(self, rank):
        dev, rb, re_, ms = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_float(0)
        check(lib.pt_mgpu_tile(self.handle, rank, ctypes.byref(dev), ctypes.byref(rb), ctypes.byref(re_), ctypes.byref(ms)))
        return {"rank": rank, "device": dev.value, "rows": (rb.value, re_.value), "kernel_ms": ms.value}

    def frame_stats(self):
        """Last frame: bands per tile, the longest rank's render ms, and what the exchange added on top (wall - render)."""
        b, r, x = ctypes.c_int(0), ctypes.c_float(0), ctypes.c_float(0)
        check(lib.pt_mgpu_frame_stats(self.handle, ctypes.byref(b), ctypes.byref(r), ctypes.byref(x)))
        return {"bands": b.value, "render_ms": r.value, "exposed_ms": x.value}

    def backend(self):
        buf = ctypes.create_string_buffer(128)
        check(lib.pt_mgpu_backend(self.handle, buf, 128))
        return buf.value.decode()

    def destroy(self):
        if self.handle:
            check(lib.pt_mgpu_destroy(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def render_frame(width, height, spp, spheres=None, basis=None, eye=DEFAULT_EYE, **opts):
    """Convenience for tests: allocate, render rows [row_begin,row_end) on the GPU, download.
    Returns (float32 [rows][width][14], kernel_ms)."""
    if spheres is None:
        spheres = scene_cornell()
    if basis is None:
        basis = camera_basis(eye, width=width, height=height)
    r = Renderer(width, height, spp, **opts)
    d_scene, n = upload_scene(spheres)
    d_out = DeviceBuffer(max(r.tile_floats * 4, 4))
    try:
        ms = r.render(d_out.ptr, d_scene.ptr, n, basis, eye)
        img = d_out.download(np.float32, (r.rows, width, CHANNELS))
    finally:
        d_out.free()
        d_scene.free()
        r.destroy()
    return img, ms
