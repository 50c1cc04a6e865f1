"""Small textured-mesh cases for the texture-bake tests (shared by the CPU and GPU suites)."""
import numpy as np


def make_case(seed, n=6000, grid=6, k=20, jitter=0.02, degenerate=False):
    """A bumpy height-field cloud over the unit square and a (grid x grid x 2)-triangle mesh under it, UV = xy scaled into
    (0.05, 0.95).  Returns (src_xyz (3,n) f64, src_rgb (n,3) u8, vert POINT_DTYPE-like dict, faces (F,3) i32)."""
    rng = np.random.default_rng(seed)
    x = rng.random(n); y = rng.random(n)
    z = 0.1 * np.sin(5 * x) * np.cos(4 * y) + jitter * rng.standard_normal(n)
    src = np.stack([x, y, z])
    if degenerate:                                   # exact duplicates and points exactly on mesh vertices / edges
        src[:, 1::7] = src[:, 0:-1:7][:, : src[:, 1::7].shape[1]]
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    g = np.linspace(0.0, 1.0, grid + 1)
    vx, vy = np.meshgrid(g, g, indexing="xy")
    vx = vx.ravel(); vy = vy.ravel()
    vz = 0.1 * np.sin(5 * vx) * np.cos(4 * vy)
    verts = np.stack([vx, vy, vz])
    if degenerate:
        src[:, :verts.shape[1]] = verts             # cloud points exactly on the mesh vertices
    uv = np.stack([0.05 + 0.9 * vx, 0.05 + 0.9 * vy], axis=1)
    vrgb = rng.integers(0, 256, size=(verts.shape[1], 3), dtype=np.uint8)
    faces = []
    for j in range(grid):
        for i in range(grid):
            a = j * (grid + 1) + i
            faces.append([a, a + 1, a + grid + 2]); faces.append([a, a + grid + 2, a + grid + 1])
    faces = np.array(faces, np.int32)
    if degenerate:
        faces = np.vstack([faces, [[0, 0, 1]], [[2, 2, 2]]]).astype(np.int32)        # zero-area faces
    return src, rgb, verts, uv, vrgb, faces


def point_records(dtype, xyz, rgb, uv=None):
    """reference Point records (80 B) from planar xyz, colours and optional UVs"""
    n = xyz.shape[1]
    a = np.zeros(n, dtype=dtype)
    a["ver"] = np.ascontiguousarray(xyz.T)
    a["color"] = rgb.astype(np.int32)
    if uv is not None:
        a["U"] = uv[:, 0]; a["V"] = uv[:, 1]
    return a
