"""The grid builder (csrc/pt_grid.h, build_grid_kernel) checked against the geometry in float64, through the lab library's
pt_debug_grid_image: what EXACTNESS.md A.6 (i)-(ii) REQUIRE of the registrations -- every cell that contains a point of a grid
sphere's (slightly inflated) ball lists that sphere -- and what round 5's ball-reach rule promises on top: nothing is registered
outside the old bounding-box rule, and the corner cells the ball does not reach are gone.  The parity tests and soaks compare
pictures; this one looks at the table itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def scenes(pt):
    rs = np.random.default_rng(11)
    out = [("random1000_walls", pt.scene_random(1000, seed=1, with_walls=True)),
           ("random1000_open", pt.scene_random(1000, seed=1, with_walls=False)),
           ("random300_walls", pt.scene_random(300, seed=3, with_walls=True))]
    sc = pt.scene_random(800, seed=5, with_walls=True)  # radii over a decade and a half: a few spheres leave the grid at either end
    sc["radius"][7:] = np.exp(rs.uniform(np.log(0.4), np.log(12.0), len(sc) - 7)).astype(np.float32)
    out.append(("radii_two_decades", sc))
    sc = pt.scene_random(600, seed=7, with_walls=False)  # far from the origin: the cell faces round coarsely
    sc["pos"] += np.float32(5000.0)
    out.append(("far_from_origin", sc))
    return out


def cell_distance(pos, lo, cs, dims):
    """float64 distance from each sphere centre to each cell's box: (n_spheres, n_cells), cells in the table's order (x fastest)"""
    nx, ny, nz = dims
    d2 = 0.0
    grids = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")  # z, y, x
    for k, idx in ((0, grids[2]), (1, grids[1]), (2, grids[0])):
        f0 = lo[k] + idx.reshape(-1).astype(np.float64) * cs
        p = pos[:, k:k + 1].astype(np.float64)
        dk = np.maximum(np.maximum(f0[None, :] - p, p - (f0[None, :] + cs)), 0.0)
        d2 = d2 + dk * dk
    return np.sqrt(d2)


@pytest.mark.parametrize("threads", [512, 1024])
def test_registrations_cover_every_cell_the_ball_reaches(lab, gpu, threads):
    for name, sc in scenes(lab):
        g = lab.grid_image(sc, eye=(50.0, 52.0, 295.6), threads=threads)
        assert g["valid"] == 1, name
        lo, cs, dims = np.asarray(g["origin"], dtype=np.float64), float(g["cell_size"]), g["dims"]
        ncells = dims[0] * dims[1] * dims[2]
        r = sc["radius"].astype(np.float64)
        in_grid = (sc["radius"] >= np.float32(g["r_small"])) & (sc["radius"] <= np.float32(g["r_big"]))
        assert sorted(g["big"].tolist()) == np.nonzero(~in_grid)[0].tolist(), name  # everything else is tested by every ray
        reg = np.zeros((len(sc), ncells), dtype=bool)
        for c, lst in enumerate(g["cells"]):
            assert len(set(lst.tolist())) == len(lst), (name, c)  # no sphere twice in a cell
            reg[lst, c] = True
            assert sorted(g["pooled"][c].tolist()) == sorted(lst.tolist()), (name, c)  # the pooled table lists the same spheres
        assert not reg[~in_grid].any(), name
        dist = cell_distance(sc["pos"], lo, cs, dims)
        # REQUIRED (A.6): the cells within r (1 + 1e-5) + slack of the centre -- the reference's float t puts its hit point within
        # 2^-21 |off|^2 / r of the surface, and the DDA may be one sliver (<< slack) off the cell
        need = (dist <= (r * (1 + 1e-5) + g["slack"])[:, None]) & in_grid[:, None]
        missing = need & ~reg
        assert not missing.any(), (name, np.argwhere(missing)[:4])
        # PROMISED (round 5): nothing beyond the ball's reach, i.e. the corners are gone.  reach = m + slack + guard with the box
        # rule's m = r + 2^-20 D^2 / r + slack, D = far + 0.875 E, slack = 2^-13 E (build_grid_kernel)
        extent = g["slack"] * 8192.0
        d_far = np.sqrt(g["far2"]) + 0.875 * extent
        m = r + 2.0 ** -20 * d_far * d_far / r + g["slack"]
        reach_up = 1.001 * (m + g["slack"] + 1e-6 * (np.abs(sc["pos"]).max() + extent))
        extra = reg & (dist > reach_up[:, None])
        assert not extra.any(), (name, np.argwhere(extra)[:4])
        if name.startswith("random1000"):
            # the bounding-box rule would have registered these cells as well: count what the ball rule saves
            p = sc["pos"].astype(np.float64)[:, None, :]
            m = m[:, None, None]
            a = np.floor((p - m - lo) / cs).clip(0, np.asarray(dims) - 1)
            b = np.floor((p + m - lo) / cs).clip(0, np.asarray(dims) - 1)
            box_count = np.prod(b - a + 1, axis=-1).reshape(-1)[in_grid].sum()
            assert reg.sum() < 0.95 * box_count, (name, int(reg.sum()), int(box_count))
