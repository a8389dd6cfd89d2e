#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.

PROVENANCE -- read before trusting these files:
  * survey_kats.json is DATA COPIED FROM SURVEY.md sections 8(a) a2 / a11 / 8(c): values the
    survey recorded from the reference's src/pathtrace.cu (host-compiled, clang++,
    -ffp-contract=off, closed-form camera basis).  They are the only reference-derived
    vectors that exist: the reference tree has no tests or golden images, and the reference
    cannot be built in this image (no CUDA / cuRAND / helper_math.h; stand-ins are not
    allowed), so nothing here was produced by running the reference in this round.
  * philox_kats.json: published Random123 known-answer vectors for Philox4x32-10.
  * oracle_*.npz are outputs of THIS REPO's CPU oracle (oracle/pt_oracle.c), committed so that
    (a) the oracle is regression-pinned and (b) the GPU tests can compare the HIP kernel with
    stored vectors as well as with the live oracle.  They are oracle-generated, NOT
    reference-generated; parity stays "unpinned" in the sense of DESIGN.md.
Run: python tests/golden/make_golden.py   (needs oracle/libpt_oracle.so; no GPU)."""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402



def main():
    O = ge.load_oracle()
    O.build()
    _generate(O)



def closed_form_basis(w, h):
    """out = z_e (F +- tan22.5 aspect R +- tan22.5 U), z_e = 2fn/(f+n)."""
    n, f = 0.01, 1000.0
    ze = 2 * f * n / (f + n)
    t = math.tan(math.radians(22.5))
    a = w / h
    F, R, U = np.array([0, 0, -1.0]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0])
    out = [ze * (F + sx * t * a * R + sy * t * U) for sx, sy in [(-1, -1), (1, -1), (-1, 1), (1, 1)]]
    return np.array(out).astype(np.float32).reshape(12)


def _generate(O):
    survey = {
        "_provenance": "copied from SURVEY.md 8(a) a2, a11 and 8(c); recorded there from the reference source "
                       "(clang++ host build, -O2 -ffp-contract=off, closed-form basis, default camera)",
        "xorwow_first3_uniforms": {
            "0": [0.740219355, 0.438451141, 0.517012656],
            "1": [0.674789965, 0.724492013, 0.507475257],
            "65535": [0.163366154, 0.957131028, 0.932381332],
        },
        "camera_basis_closed_form_B0": [-0.00828418881, -0.00828418881, -0.0199998002],
        "aov_256_spp1": {
            "128,128": {"normal": [0.0, -0.000112, 1.0], "albedo": [0.75, 0.75, 0.75], "depth": 14780.2},
            "0,0": {"normal": [0.000294956, -0.999999, -0.00142791], "albedo": [0.75, 0.75, 0.75], "depth": 3560.47},
            "255,255": {"normal": [-1.0, 0.000377472, -0.000949005], "albedo": [0.25, 0.25, 0.75], "depth": 5955.04},
            "192,64": {"normal": [1.0, 0.000377897, 0.000225432], "albedo": [0.75, 0.25, 0.25], "depth": 11827.3},
        },
        "means_256_spp4": {
            "color_clang": [0.170195, 0.114189, 0.137857],
            "color_gxx": [0.171258, 0.115258, 0.138657],
            "normal": [0.00366436, -0.145765, 0.134215],
            "albedo": [0.666182, 0.578234, 0.664026],
            "depth": 8426.65,
            "colorvar_clang": 0.197727,
            "colorvar_gxx": 0.199144,
            "normalvar": 0.00125747,
            "albedovar": 0.000556624,
            "depthvar": 5719.06,
        },
        "exr_8x8_ramp": {"bytes": 4327, "scanline0_plane0": [8, 22, 36, 50, 64, 78, 92, 106]},
    }
    json.dump(survey, open(os.path.join(HERE, "survey_kats.json"), "w"), indent=1)

    philox = {
        "_provenance": "Random123 kat_vectors (Salmon et al., SC'11), philox4x32 10 rounds",
        "vectors": [
            {"ctr": [0, 0, 0, 0], "key": [0, 0], "out": [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]},
            {"ctr": [0xFFFFFFFF] * 4, "key": [0xFFFFFFFF] * 2, "out": [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]},
            {"ctr": [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], "key": [0xA4093822, 0x299F31D0],
             "out": [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]},
        ],
    }
    json.dump(philox, open(os.path.join(HERE, "philox_kats.json"), "w"), indent=1)

    cornell = O.scene_cornell()
    eye = np.array([50.0, 52.0, 295.6], dtype=np.float32)


    def save(name, size, spp, rng, basis, rows=None, **kw):
        img = O.render(size, size, spp, spheres=cornell, basis=basis, eye=eye, rng_mode=rng, threads=8, **kw)
        means = img.reshape(-1, 14).mean(0, dtype=np.float64)
        data = {"basis": basis, "eye": eye, "size": size, "spp": spp, "rng": rng, "means": means,
                "min": img.reshape(-1, 14).min(0), "max": img.reshape(-1, 14).max(0)}
        if rows is None:
            data["image"] = img
        else:
            data["rows"] = np.array(rows)
            data["row_data"] = img[rows]
        np.savez_compressed(os.path.join(HERE, name), **data)
        print(name, "means", means[:3], "size", os.path.getsize(os.path.join(HERE, name)))


    # F1/F2: 64x64 full buffers, both generators, closed-form basis (the SURVEY's setup)
    b64 = closed_form_basis(64, 64)
    save("oracle_64_spp1_xorwow.npz", 64, 1, 0, b64)
    save("oracle_64_spp4_xorwow.npz", 64, 4, 0, b64)
    save("oracle_64_spp4_philox.npz", 64, 4, 1, b64)
    # F3: config 1 (256x256x4): stats + 4 rows
    b256 = closed_form_basis(256, 256)
    save("oracle_256_spp4_xorwow_rows.npz", 256, 4, 0, b256, rows=[0, 128, 192, 255])
    # glm-pipeline basis (what the product's Camera look-alike produces) at 64x64x16, 8 bounces (config 5 shape)
    bg = O.camera_basis(w=64, h=64)
    save("oracle_64_spp16_xorwow_glm_b8.npz", 64, 16, 0, bg, max_bounces=8)


if __name__ == "__main__":
    main()
