"""Furnace known-answer shared by the CPU (oracle) and GPU (HIP path) tests: the four reference-rendered furnace images whose
material is unambiguous - and, round 3, the centre of a fifth that pins the ROUGH GGX lobe - (tests/golden/furnace_reference.json, built by tests/golden/make_furnace_fixture.py from
thesis/assets/furnace-test/*.png of the reference) against our render of the same setup.

The reference's sphere geometry is a missing blob (assets/sphere.obj.scene); only its silhouette is known from the images
(half-height 302 px of 1024 at vfov 50 from 3 units away => radius 0.795 around (0, 1, 0)).  A convex object in a uniform white
environment shows, at every pixel, the directional albedo of its material at that pixel's viewing angle, so the comparison is made
ring by ring in radius normalised to the silhouette, on the 8-bit values the reference wrote (owl::make_rgba, device.cu:252-253).

Tolerance (stated): ring means within 1.0 code value of the reference's (its spp is unknown; at our 256 spp the ring means carry
< 0.1 code of Monte-Carlo noise), fully saturated images stay saturated, near-saturated ones keep >= 98.5 % of the sphere at 255.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = json.load(open(os.path.join(HERE, "golden", "furnace_reference.json")))
RADIUS = 0.795
W = H = 256
SPP = 256
DEPTH = 16


def setup(scene_io, procedural, key):
    base = dict(base_color=[1, 1, 1], specular=0.0, specular_tint=0.0, roughness=1.0, sheen_tint=0.0, clearcoat_gloss=0.0, ior=1.5)  # assets/sphere.json
    base.update(FIXTURE["images"][key]["material_overrides"])
    mat = scene_io.material(**base)
    mats = [("sphere", mat, "")]
    ents = scene_io.build_entities([("sphere", procedural.uv_sphere((0, 1, 0), RADIUS, nu=192, nv=96))], mats)
    return ents, mats


def ring_means(rgba8):
    g = (np.asarray(rgba8) & 0xFF).astype(np.float64)
    assert ((np.asarray(rgba8) >> 8) & 0xFF == g).all() and ((np.asarray(rgba8) >> 16) & 0xFF == g).all()  # grey
    yy, xx = np.mgrid[0:H, 0:W]
    rr = np.sqrt((yy - (H - 1) / 2.0) ** 2 + (xx - (W - 1) / 2.0) ** 2) / (FIXTURE["silhouette_radius_over_half_image"] * (H / 2.0))
    return [float(g[(rr >= a) & (rr < a + 0.1)].mean()) for a in np.arange(0, 1.0, 0.1)], g, rr


def check(key, rgba8):
    ref = FIXTURE["images"][key]
    rings, g, rr = ring_means(rgba8)
    inside = rr < 0.97
    if ref["min"] == 255.0:  # diffuse_roughness(1.0): the reference image is 255 everywhere
        assert (g == 255).all(), "%s: %d pixels below 255" % (key, int((g < 255).sum()))
        return rings
    if "pinned_rings" in ref:
        # metallic_vndf_roughness(1.0): rendered by the thesis' VNDF-SAMPLED variant of the rough GGX lobe; the shipped lobe (NDF-sampled,
        # VNDF pdf) must agree with it only at normal incidence (tests/golden/make_furnace_fixture.py has the argument), i.e. in the
        # centre rings.  This pins D_GTR2, G2, lambda and the Fresnel term at alpha = 1 with reference-rendered data.
        for k in ref["pinned_rings"]:
            assert abs(rings[k] - ref["ring_means"][k]) <= 1.0, (key, k, rings[k], ref["ring_means"][k])
        assert rings[-1] < ref["ring_means"][-1]  # and the rim is darker than the VNDF-sampled image, as the weights predict
        return rings
    assert np.abs(np.array(rings) - np.array(ref["ring_means"])).max() <= 1.0, (key, rings, ref["ring_means"])
    if ref["fraction_255"] > 0.99:  # mirror-like: saturated except at the very rim
        assert (g[inside] == 255).mean() >= 0.985, (key, float((g[inside] == 255).mean()))
    else:
        c = g[H // 2 - 5:H // 2 + 5, W // 2 - 5:W // 2 + 5]
        assert abs(c.mean() - ref["centre_mean"]) <= 1.0, (key, c.mean(), ref["centre_mean"])
    return rings
