"""A SECOND restatement of the reference's per-pixel path (src/pathtrace.cu:39-257) -- vectorised numpy over the pixels of a frame,
one ufunc per IEEE operation with explicit float32 / float64 / uint32 dtypes -- written from the reference's lines and sharing no
code with oracle/pt_oracle.c or the HIP kernels.  Test infrastructure (tests/test_first_hit_numpy.py, tests/test_full_path_numpy.py).

What it takes from the CONTRACT rather than from the reference (oracle/pt_oracle.c header), because the reference leaves it to
closed CUDA libraries: C2 rsqrtf := 1/sqrtf, C3 powf(u, 0.5f) := sqrtf(u), C5 the draw order of :131, C8 the XORWOW constants
(restated here in uint32 arithmetic), and C4 -- sin / cos of :135 are the contract's own polynomial, whose explicit fmaf has no
numpy counterpart: `sincos` is passed in by the caller (the tests pass the oracle's pto_sincos).  Everything else -- promotions,
operation order, the scene loop, the bounce loop, the accumulators and Welford updates -- is read off the reference here."""
import numpy as np

f32, f64, u32 = np.float32, np.float64, np.uint32
EYE = (50.0, 52.0, 295.6)
PUSH_RAY_ORIGIN = f32(0.05)       # :8
PI_F = f32(3.141592654)           # CUDART_PI_F (math_constants.h)


def _dot(a, b):  # helper_math dot(float3, float3): a.x*b.x + a.y*b.y + a.z*b.z, float, left to right
    return ((a[0] * b[0]).astype(f32) + (a[1] * b[1]).astype(f32)).astype(f32) + (a[2] * b[2]).astype(f32)


def _lerp(a, b, t):  # helper_math lerp: a + t*(b-a)
    return [(a[k] + (t * (b[k] - a[k]).astype(f32)).astype(f32)).astype(f32) for k in range(3)]


def _normalize(v):  # helper_math normalize: v * rsqrtf(dot(v, v)); contract C2
    with np.errstate(invalid="ignore", divide="ignore"):
        inv = (f32(1.0) / np.sqrt(_dot(v, v)).astype(f32)).astype(f32)
        return [(v[k] * inv).astype(f32) for k in range(3)]


def _cross(a, b):  # helper_math cross
    return [((a[1] * b[2]).astype(f32) - (a[2] * b[1]).astype(f32)).astype(f32),
            ((a[2] * b[0]).astype(f32) - (a[0] * b[2]).astype(f32)).astype(f32),
            ((a[0] * b[1]).astype(f32) - (a[1] * b[0]).astype(f32)).astype(f32)]


def _luminance(c):  # :67-69  double literals: the whole expression is double, returned as float
    return (f64(0.2126) * c[0].astype(f64) + f64(0.7152) * c[1].astype(f64) + f64(0.0722) * c[2].astype(f64)).astype(f32)


def intersect_scene(o, d, spheres, promote=True):
    """intersectScene / intersectSphere (:72-107) for arrays of rays: (hit, t of the hit, index).
    promote=False: a deliberately WRONG reading -- :80-81 evaluated in float -- for the sensitivity test."""
    wide = f64 if promote else f32
    shape = o[0].shape
    t_nearest = np.full(shape, f32(1000000.0))                                   # :94
    t = np.zeros(shape, dtype=f32)                                               # :95 (persists across the spheres)
    hit = np.zeros(shape, dtype=bool)
    index = np.zeros(shape, dtype=np.int64)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        a = _dot(d, d)                                                           # :74 (the same for every sphere)
        for i, s in enumerate(spheres):                                          # :97
            pos = [f32(s["pos"][k]) for k in range(3)]
            r = f32(s["radius"])
            off = [(o[k] - pos[k]).astype(f32) for k in range(3)]                 # :73
            b = (f64(2.0) * _dot(d, off).astype(f64)).astype(f32)                 # :75  double product, stored in a float
            c = (_dot(off, off) - (r * r).astype(f32)).astype(f32)                # :76
            det = ((b * b).astype(f32) - ((f32(4.0) * a).astype(f32) * c).astype(f32)).astype(f32)  # :77  int 4 -> float
            inside = det >= 0                                                    # :79
            bb = (b * b).astype(f32).astype(wide)                                # :80  (b*b) is a float product
            root = np.sqrt(bb - (wide(4.0) * a.astype(wide)) * c.astype(wide))   # 4.0*a*c = (4.0*a)*c in double
            den = wide(2.0) * a.astype(wide)
            t_near = (((-b).astype(wide) - root) / den).astype(f32)              # :80
            t_far = (((-b).astype(wide) + root) / den).astype(f32)               # :81
            both = (t_near > 0) & (t_far > 0)
            t_new = np.where(both, np.minimum(t_near, t_far), np.where(t_near > 0, t_near, t_far))  # :82-87
            t = np.where(inside, t_new, t)                                       # *t written only on a hit
            take = inside & (t > 0) & (t < t_nearest)                            # :99
            t_nearest = np.where(take, t, t_nearest)
            index = np.where(take, i, index)
            hit |= take
    return hit, t_nearest, index


def _hit_geometry(o, d, th, index, spheres):
    centre = np.asarray([s["pos"] for s in spheres], dtype=f32)[index]
    p = [(o[k] + (d[k] * th).astype(f32)).astype(f32) for k in range(3)]          # :163
    n = _normalize([(p[k] - centre[..., k]).astype(f32) for k in range(3)])       # :164
    with np.errstate(invalid="ignore"):
        flip = ~(_dot(n, d) < 0)                                                 # :166
    n = [np.where(flip, (f32(-1.0) * n[k]).astype(f32), n[k]) for k in range(3)]
    return p, n


def _primary(width, height, basis, eye, jitter=None):
    B = np.asarray(basis, dtype=f32).reshape(4, 3)
    x = np.arange(height, dtype=f32)[:, None] * np.ones((1, width), dtype=f32)   # :204 (the row)
    y = np.ones((height, 1), dtype=f32) * np.arange(width, dtype=f32)[None, :]   # :205
    if jitter is not None:                                                       # :223-224
        x = (x + ((jitter[0] * f32(1.0)).astype(f32) - f32(0.5)).astype(f32)).astype(f32)
        y = (y + ((jitter[1] * f32(1.0)).astype(f32) - f32(0.5)).astype(f32)).astype(f32)
    sx = (x / f32(width)).astype(f32)                                            # :226
    sy = (y / f32(height)).astype(f32)
    b0, b1, b2, b3 = ([np.full_like(sx, B[j, k]) for k in range(3)] for j in range(4))
    d = _lerp(_lerp(b0, b1, sy), _lerp(b2, b3, sy), (f32(1.0) - sx).astype(f32))  # :229
    o = [np.full_like(sx, f32(eye[k])) for k in range(3)]
    return o, d


def first_hit_frame(width, height, spheres, basis, eye=EYE, promote=True):
    """Channels 3..9 of the reference's 1-spp frame: (height, width, 7) float32.  Square frames only (main.cu:66-67)."""
    assert width == height
    o, d = _primary(width, height, basis, eye)
    hit, th, index = intersect_scene(o, d, spheres, promote)
    colour = np.asarray([s["color"] for s in spheres], dtype=f32)[index]
    _, n = _hit_geometry(o, d, th, index, spheres)
    out = np.zeros(hit.shape + (7,), dtype=f32)
    zero = np.zeros(hit.shape, dtype=f32)
    for k in range(3):
        out[..., k] = np.where(hit, (zero + n[k]).astype(f32), zero)             # :188  L.normal += normal  (0 + -0 = +0)
        out[..., 3 + k] = np.where(hit, (zero + colour[..., k]).astype(f32), zero)   # :189
    out[..., 6] = np.where(hit, (zero + th).astype(f32), zero)                    # :190
    return out                                                                   # (:234-237: divided by (float)1)


# ---- cuRAND XORWOW (contract C8: curand_init(seed, 0, 0), curand, curand_uniform), in uint32 arithmetic on arrays ---------------
class Xorwow:
    def __init__(self, seed):
        seed = np.asarray(seed, dtype=np.uint64)
        with np.errstate(over="ignore"):
            s0 = (seed & np.uint64(0xFFFFFFFF)).astype(u32) ^ u32(0xAAD26B49)
            s1 = (seed >> np.uint64(32)).astype(u32) ^ u32(0xF7DCEFDD)
            t0 = u32(1099087573) * s0
            t1 = u32(2591861531) * s1
            self.d = u32(6615241) + t1 + t0
            self.v = [u32(123456789) + t0, u32(362436069) ^ t0, u32(521288629) + t1, u32(88675123) ^ t1, u32(5783321) + t0]

    def uniform(self, where):
        """One curand_uniform for the pixels in `where` (the others keep their state); float32 in (0, 1]."""
        with np.errstate(over="ignore"):
            t = self.v[0] ^ (self.v[0] >> u32(2))
            v4 = (self.v[4] ^ (self.v[4] << u32(4))) ^ (t ^ (t << u32(1)))
            new_v = [self.v[1], self.v[2], self.v[3], self.v[4], v4]
            new_d = self.d + u32(362437)
            x = v4 + new_d
        self.v = [np.where(where, new_v[k], self.v[k]) for k in range(5)]
        self.d = np.where(where, new_d, self.d)
        c = f32(2.3283064e-10)
        return ((x.astype(f32) * c).astype(f32) + (c / f32(2.0)).astype(f32)).astype(f32)


def _welford(state, x, where):  # :52-58  n int, mean / M2 float; delta / n is float / (float)n
    n, mean, m2 = state
    n1 = n + 1
    delta = (x - mean).astype(f32)
    with np.errstate(invalid="ignore", over="ignore"):
        mean1 = (mean + (delta / n1.astype(f32)).astype(f32)).astype(f32)
        delta2 = (x - mean1).astype(f32)
        m21 = (m2 + (delta * delta2).astype(f32)).astype(f32)
    return np.where(where, n1, n), np.where(where, mean1, mean), np.where(where, m21, m2)


def _variance(state):  # :60-64
    n, _, m2 = state
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(n < 2, f32(0.0), (m2 / (n - 1).astype(f32)).astype(f32)).astype(f32)


def render_frame(width, height, spp, spheres, basis, sincos, eye=EYE, max_bounces=5, seed=0):
    """pixel_kernel + trace_ray (:150-257) with the XORWOW stream of setup_random (:259-266: seed = pixel id): (h, w, 14) float32.
    `sincos(x_array) -> (sin_array, cos_array)` is contract C4's definition."""
    assert width == height
    shape = (height, width)
    ids = (np.arange(height, dtype=np.uint64)[:, None] * np.uint64(width) + np.arange(width, dtype=np.uint64)[None, :]) + np.uint64(seed)
    rng = Xorwow(ids)                                                            # :265
    everyone = np.ones(shape, dtype=bool)
    emission = np.asarray([s["emission"] for s in spheres], dtype=f32)
    colour_of = np.asarray([s["color"] for s in spheres], dtype=f32)
    zero = np.zeros(shape, dtype=f32)
    L_color, L_normal, L_albedo = [zero.copy() for _ in range(3)], [zero.copy() for _ in range(3)], [zero.copy() for _ in range(3)]
    L_depth = zero.copy()
    var = [(np.zeros(shape, dtype=np.int32), zero.copy(), zero.copy()) for _ in range(4)]   # COLOR, NORMAL, ALBEDO, DEPTH (:21)

    for _ in range(spp):                                                         # :218
        jitter = None
        if spp != 1:                                                             # :222
            jx = rng.uniform(everyone)
            jy = rng.uniform(everyone)
            jitter = (jx, jy)
        o, d = _primary(width, height, basis, eye, jitter)
        color = [zero.copy() for _ in range(3)]
        mask = [np.ones(shape, dtype=f32) for _ in range(3)]
        alive = everyone.copy()
        for n in range(max_bounces):                                             # :155
            hit, th, index = intersect_scene(o, d, spheres)
            gone = alive & ~hit                                                  # :157-161: colour added, NO variance update, return
            for k in range(3):
                L_color[k] = np.where(gone, (L_color[k] + color[k]).astype(f32), L_color[k])
            alive = alive & hit
            p, nrm = _hit_geometry(o, d, th, index, spheres)
            em, col = emission[index], colour_of[index]
            for k in range(3):
                me = (mask[k] * em[..., k]).astype(f32)
                if n == 0:                                                       # :171-172  clamp(v, 0, 1) = fmaxf(0, fminf(v, 1))
                    me = np.fmax(f32(0.0), np.fmin(me, f32(1.0)))
                color[k] = np.where(alive, (color[k] + me).astype(f32), color[k])   # :172 / :174
                mask[k] = np.where(alive, (mask[k] * col[..., k]).astype(f32), mask[k])  # :175
            o_next = [(p[k] + (nrm[k] * PUSH_RAY_ORIGIN).astype(f32)).astype(f32) for k in range(3)]   # :178
            # getCosineWeightedNormal(normal, randState) :126-136
            dirn = _normalize(nrm)                                               # :128
            with np.errstate(invalid="ignore"):
                pick = np.abs(dirn[0]) > np.abs(dirn[2])                          # :123
            neg = lambda v: (-v).astype(f32)  # noqa: E731
            ortho = [np.where(pick, neg(dirn[1]), zero), np.where(pick, dirn[0], neg(dirn[2])), np.where(pick, zero, dirn[1])]
            o1 = _normalize(ortho)                                               # :129
            o2 = _normalize(_cross(dirn, o1))                                    # :130
            u_az = rng.uniform(alive)                                            # :131, first draw -> r.x (contract C5)
            u_el = rng.uniform(alive)
            rx = ((u_az * f32(2.0)).astype(f32) * PI_F).astype(f32)              # :132
            ry = np.sqrt(u_el).astype(f32)                                       # :133  pow(r.y, 0.5f) := sqrtf (C3)
            with np.errstate(invalid="ignore"):
                oneminus = np.sqrt(f64(1.0) - (ry * ry).astype(f32).astype(f64)).astype(f32)   # :134  double, stored in a float
            sn, cs = sincos(rx)
            ca, sa = (cs * oneminus).astype(f32), (sn * oneminus).astype(f32)
            nd = [(((ca * o1[k]).astype(f32) + (sa * o2[k]).astype(f32)).astype(f32) + (ry * dirn[k]).astype(f32)).astype(f32) for k in range(3)]  # :135
            d_next = _normalize(nd)                                              # :180
            if n == 0:                                                           # :187-195
                lum_n, lum_a = _luminance(nrm), _luminance([col[..., k] for k in range(3)])
                for k in range(3):
                    L_normal[k] = np.where(alive, (L_normal[k] + nrm[k]).astype(f32), L_normal[k])
                    L_albedo[k] = np.where(alive, (L_albedo[k] + col[..., k]).astype(f32), L_albedo[k])
                L_depth = np.where(alive, (L_depth + th).astype(f32), L_depth)
                var[1] = _welford(var[1], lum_n, alive)
                var[2] = _welford(var[2], lum_a, alive)
                var[3] = _welford(var[3], th, alive)
            o = [np.where(alive, o_next[k], o[k]) for k in range(3)]
            d = [np.where(alive, d_next[k], d[k]) for k in range(3)]
        for k in range(3):                                                       # :198
            L_color[k] = np.where(alive, (L_color[k] + color[k]).astype(f32), L_color[k])
        var[0] = _welford(var[0], _luminance(color), alive)                      # :200

    out = np.zeros(shape + (14,), dtype=f32)
    nf = f32(spp)
    for k in range(3):                                                           # :234-236
        out[..., k] = (L_color[k] / nf).astype(f32)
        out[..., 3 + k] = (L_normal[k] / nf).astype(f32)
        out[..., 6 + k] = (L_albedo[k] / nf).astype(f32)
    out[..., 9] = (L_depth / nf).astype(f32)
    for f in range(4):                                                           # :251-254
        out[..., 10 + f] = _variance(var[f])
    return out
