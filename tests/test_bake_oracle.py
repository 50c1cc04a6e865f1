"""CPU checks of the texture-bake oracle (oracle/pt_oracle.c: pto_bake_texture, pto_dilate_pad) -- the build's definition of
the reference's per-face bake (src/pointsTransfer.cpp:462-581, :66-107) and edge padding (:593-611) -- against independent
restatements: a numpy rasteriser for the no-neighbour case, scipy's Delaunay for the number and total area of sub-triangles
(through coverage), scipy.ndimage for the dilation."""
import numpy as np
import pytest

from _bake_cases import make_case


def _numpy_draw(tex, U, V, col, R):
    """reference draw_triangle (:66-107) in numpy, same operation order as the oracle"""
    p = np.array([U[0] * R, V[0] * R]); q = np.array([U[1] * R, V[1] * R]); r = np.array([U[2] * R, V[2] * R])
    A = (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])
    if A == 0:
        return
    i0 = max(int(np.floor(min(p[0], q[0], r[0]))), 0); i1 = min(int(np.floor(max(p[0], q[0], r[0]))), R - 1)
    j0 = max(int(np.floor(min(p[1], q[1], r[1]))), 1); j1 = min(int(np.floor(max(p[1], q[1], r[1]))), R)
    for i in range(i0, i1 + 1):
        for j in range(j0, j1 + 1):
            x, y = float(min(i, R - 1)), float(min(j, R - 1))
            b0 = ((q[0] - x) * (r[1] - y) - (q[1] - y) * (r[0] - x)) / A
            b1 = ((r[0] - x) * (p[1] - y) - (r[1] - y) * (p[0] - x)) / A
            b2 = (1.0 - b0) - b1
            if b0 >= 0 and b1 >= 0 and b2 >= 0:
                for c in range(3):
                    f = np.float32((b0 * float(col[0][c]) + b1 * float(col[1][c])) + b2 * float(col[2][c]))
                    tex[R - j, i, 2 - c] = np.uint8(min(max(f, np.float32(0)), np.float32(255)))
                tex[R - j, i, 3] = 255


def test_bake_without_neighbours_is_the_plain_rasteriser(oracle):
    src, rgb, verts, uv, vrgb, faces = make_case(3, n=50, grid=3)
    R = 96
    none = np.full((verts.shape[1], 4), 0xFFFFFFFF, np.uint32)
    got = oracle.bake_texture(src, rgb, verts, uv, vrgb, faces, none, R)
    want = np.zeros((R, R, 4), np.uint8)
    for f in faces:
        _numpy_draw(want, uv[f, 0], uv[f, 1], vrgb[f], R)
    assert np.array_equal(got, want)


def test_bake_covers_the_face_and_uses_cloud_colours(oracle):
    """With neighbours, the sub-triangles tile each face: whatever the plain face rasterisation covers stays covered, except
    pixel centres that sit exactly on an edge (barycentric rounding, as in the reference) -- and the colours now come from the cloud."""
    src, rgb, verts, uv, vrgb, faces = make_case(4, n=4000, grid=4)
    uv = uv * np.array([0.987, 0.981]) + np.array([0.0031, 0.0057])      # keep mesh edges (the diagonals too) off the pixel centres
    R = 256
    idx, _ = oracle.knn_bruteforce(src, verts, 20)
    none = np.full_like(idx, 0xFFFFFFFF)
    plain = oracle.bake_texture(src, rgb, verts, uv, vrgb, faces, none, R)
    baked = oracle.bake_texture(src, rgb, verts, uv, vrgb, faces, idx, R)
    lost = (plain[:, :, 3] == 255) & (baked[:, :, 3] != 255)
    assert (plain[:, :, 3] == 255).sum() > 40000 and lost.sum() <= 4
    assert (baked[:, :, 3] == 255).sum() <= (plain[:, :, 3] == 255).sum()          # interior points never reach outside their face
    assert (baked[:, :, :3] != plain[:, :, :3]).any(axis=2).mean() > 0.3
    # idempotent and order-defined: baking twice gives the same bytes
    assert np.array_equal(baked, oracle.bake_texture(src, rgb, verts, uv, vrgb, faces, idx, R))


def test_bake_degenerate_inputs(oracle):
    """duplicate cloud points, cloud points exactly on mesh vertices, zero-area faces, out-of-range indices, UVs outside [0, 1]"""
    src, rgb, verts, uv, vrgb, faces = make_case(5, n=3000, grid=3, degenerate=True)
    idx, _ = oracle.knn_bruteforce(src, verts, 20)
    idx[3, :5] = 0xFFFFFFFF
    uv2 = uv * 1.3 - 0.15                          # part of the atlas outside the texture: those pixels are skipped
    faces = np.vstack([faces, [[0, 1, 9999]]]).astype(np.int32)      # malformed face: skipped
    tex = oracle.bake_texture(src, rgb, verts, uv2, vrgb, faces, idx, 128)
    assert tex.shape == (128, 128, 4) and set(np.unique(tex[:, :, 3])) <= {0, 255}
    assert (tex[:, :, 3] == 255).mean() > 0.5


def test_dilate_pad_against_scipy(oracle):
    from scipy import ndimage
    rng = np.random.default_rng(6)
    R = 97
    tex = np.zeros((R, R, 4), np.uint8)
    m = rng.random((R, R)) < 0.02
    tex[m, :3] = rng.integers(0, 256, size=(int(m.sum()), 3), dtype=np.uint8)
    tex[m, 3] = 255
    tex[5, 5] = (10, 20, 30, 128)                   # a partly transparent pixel: the mask is bitwise, the add saturates
    for ks in (1, 3, 25):
        got = oracle.dilate_pad(tex, ks)
        dil = np.stack([ndimage.maximum_filter(tex[:, :, c], size=ks, mode="constant", cval=0) for c in range(4)], axis=2)
        mask = (~tex[:, :, 3])[:, :, None]
        want = np.minimum(tex.astype(np.int32) + (dil & mask).astype(np.int32), 255).astype(np.uint8)
        assert np.array_equal(got, want), ks
