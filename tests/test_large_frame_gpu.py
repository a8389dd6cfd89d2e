"""The largest frames the reference itself supports: its pixel index is a 32-bit int (src/pathtrace.cu:206,240: x * width * 14 + ...
overflows beyond about 12 385 x 12 385), so 9216 x 9216 -- 84.9 M pixels, a 4.76 GB frame whose byte offsets pass 4 GiB, 2 GB of
generator state -- is inside its range and must be inside ours: whole-frame properties, rows 0 / 4607 / 9215 against the
oracle, and the same frame rendered as four row tiles (what four GPUs would render), compared band by band."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZE = 9216
BAND = 256  # rows per host round trip (132 MB)


def _band(pt, buf, row, rows):
    out = np.empty((rows, SIZE, 14), dtype=np.float32)
    pt.check(pt.lib.pt_memcpy_d2h(out.ctypes.data, ctypes.c_void_p(buf.ptr + row * SIZE * 56), out.nbytes))
    return out


def test_9216_square_frame_rows_and_tiles(pt, oracle, gpu):
    basis = pt.camera_basis(width=SIZE, height=SIZE)
    scene = pt.scene_cornell()
    d_scene, n = pt.upload_scene(scene)
    nbytes = SIZE * SIZE * 56
    assert nbytes > 1 << 32
    try:
        full, tiles = pt.DeviceBuffer(nbytes), pt.DeviceBuffer(nbytes)
    except pt.PtError:
        pytest.skip("needs 2 x 4.76 GB of free HBM")
    pt.check(pt.lib.pt_memset(full.ptr, 0xFF, nbytes))
    pt.check(pt.lib.pt_memset(tiles.ptr, 0xEE, nbytes))
    r = pt.Renderer(SIZE, SIZE, 1)
    ms = r.render(full.ptr, d_scene.ptr, n, basis)
    r.destroy()
    # the same frame as four row tiles, each by its own renderer (its d_out = the first float of its first row)
    for g in range(4):
        b, e = g * SIZE // 4, (g + 1) * SIZE // 4
        rt = pt.Renderer(SIZE, SIZE, 1, row_begin=b, row_end=e)
        rt.render(ctypes.c_void_p(tiles.ptr + b * SIZE * 56), d_scene.ptr, n, basis)
        rt.destroy()
    colours = np.unique(scene["color"], axis=0)
    for row in range(0, SIZE, BAND):
        a, t = _band(pt, full, row, BAND), _band(pt, tiles, row, BAND)
        assert np.array_equal(a.view(np.uint32), t.view(np.uint32)), f"tiles differ from the frame in rows {row}..{row + BAND}"
        # whole-frame properties (1 spp: no jitter, variances are zero, every pixel sees a surface of the closed box)
        assert np.isfinite(a).all() and (a[..., 10:] == 0).all() and (a[..., 9] > 0).all()
        alb = a[..., 6:9].reshape(-1, 3)
        assert (alb[:, None, :] == colours[None, :, :]).all(-1).any(-1).all()  # every albedo is one of the scene's colours
        nn = (a[..., 3:6].astype(np.float64) ** 2).sum(-1)
        assert (np.abs(nn - 1) < 1e-5).all()                                   # unit normals
    # rows against the oracle, bit for bit (the last row's floats sit beyond byte offset 4 GiB)
    for row in (0, SIZE // 2 - 1, SIZE - 1):
        ref = oracle.render(SIZE, SIZE, 1, spheres=scene, basis=basis, row_begin=row, row_end=row + 1)
        got = _band(pt, full, row, 1)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"row {row}"
    full.free()
    tiles.free()
    print(f"9216^2 x 1 spp: {ms:.2f} ms")
