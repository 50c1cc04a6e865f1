"""ctypes front-end of the CPU oracle (oracle/pt_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.  Parity status:
"parity unpinned" for the k-NN search (see the header of pt_oracle.c); the Point
layout is pinned by oracle/_ref/point_layout.json.

All xyz arrays are planar float64 of shape (3, n).
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libpt_oracle.so")
NOIDX = 0xFFFFFFFF

_lib = None


def build(force=False):
    """Compile the oracle (gcc) and, if /root/reference exists, the layout probe."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "pt_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_build/libpt_oracle.so"])
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        u64, i32, p = C.c_uint64, C.c_int, C.c_void_p
        L.pto_splitmix64.restype = u64
        L.pto_splitmix64.argtypes = [u64]
        L.pto_synth_xyz_f32.argtypes = [u64, u64, u64, u64, p]
        L.pto_synth_xyz_dist.argtypes = [u64, u64, i32, u64, u64, u64, u64, p]
        L.pto_synth_rgb.argtypes = [u64, u64, u64, p]
        L.pto_synth_filter_boxes.restype = C.c_int64
        L.pto_synth_filter_boxes.argtypes = [u64, u64, u64, i32, p, p, u64, p, p, p]
        L.pto_synth_nrm.argtypes = [u64, u64, u64, p]
        for f in ("pto_transformed_distance",):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [p, p]
        for f in ("pto_min_distance_to_rectangle", "pto_max_distance_to_rectangle"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [p, p, p, p]
        L.pto_new_distance.restype = C.c_double
        L.pto_new_distance.argtypes = [C.c_double] * 3
        L.pto_transformed_distance_scalar.restype = C.c_double
        L.pto_transformed_distance_scalar.argtypes = [C.c_double]
        L.pto_inverse_of_transformed_distance.restype = C.c_double
        L.pto_inverse_of_transformed_distance.argtypes = [C.c_double]
        L.pto_knn_bruteforce.argtypes = [p, u64, p, p, u64, i32, p, p, i32]
        L.pto_kdtree_build.restype = p
        L.pto_kdtree_build.argtypes = [p, u64, p]
        L.pto_kdtree_free.argtypes = [p]
        L.pto_kdtree_query.argtypes = [p, p, u64, i32, p, p, i32]
        L.pto_num_threads.restype = i32
        L.pto_merge_candidates.argtypes = [p, p, i32, u64, i32, p, p]
        L.pto_blend.argtypes = [p, p, u64, i32, i32, p, p, p, p]
        L.pto_blend_weighted.argtypes = [p, p, u64, i32, p, p, p, p]
        L.pto_pca_normals.argtypes = [p, u64, i32, p, u64, p, p, p]
        L.pto_bake_texture.argtypes = [p, p, u64, p, p, p, u64, p, u64, p, i32, i32, p]
        L.pto_dilate_pad.argtypes = [p, i32, i32, p]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _planar64(xyz):
    a = np.ascontiguousarray(np.asarray(xyz, dtype=np.float64))
    assert a.ndim == 2 and a.shape[0] == 3, "xyz must be planar (3, n)"
    return a


# ---- generator -------------------------------------
def synth_xyz(seed, stream, n, i0=0, dist=0, n_total=0, m_total=0):
    """dist 0 uniform, 1 clustered (targets, stream 1, need the total source / target counts)."""
    out = np.empty((3, n), dtype=np.float32)
    if dist == 0:
        lib().pto_synth_xyz_f32(seed, stream, i0, n, _ptr(out))
    else:
        lib().pto_synth_xyz_dist(seed, stream, dist, i0, n, n_total or n, m_total or n, _ptr(out))
    return out


def synth_filter_boxes(seed, stream, n_total, lo, hi, cap=4_000_000):
    """Stream the uniform generator over [0, n_total) and keep the points inside any of the boxes [lo[b], hi[b]) (float32
    (nbox, 3) arrays): returns (xyz (3, c) float32, original indices (c,) uint32, box of each point (c,) int32), ascending index."""
    lo = np.ascontiguousarray(lo, dtype=np.float32); hi = np.ascontiguousarray(hi, dtype=np.float32)
    assert lo.shape == hi.shape and lo.ndim == 2 and lo.shape[1] == 3
    xyz = np.empty((cap, 3), dtype=np.float32); idx = np.empty(cap, dtype=np.uint32); box = np.empty(cap, dtype=np.int32)
    c = lib().pto_synth_filter_boxes(seed, stream, n_total, lo.shape[0], _ptr(lo), _ptr(hi), cap, _ptr(xyz), _ptr(idx), _ptr(box))
    assert 0 <= c <= cap, "box filter: %d points for a capacity of %d" % (c, cap)
    return np.ascontiguousarray(xyz[:c].T), idx[:c].copy(), box[:c].copy()


def synth_rgb(seed, n, i0=0):
    out = np.empty((n, 3), dtype=np.uint8)
    lib().pto_synth_rgb(seed, i0, n, _ptr(out))
    return out


def synth_nrm(seed, n, i0=0):
    out = np.empty((n, 3), dtype=np.float32)
    lib().pto_synth_nrm(seed, i0, n, _ptr(out))
    return out


# ---- Distance.h restatement -------------------------------------------------
def transformed_distance(p, q):
    p = np.asarray(p, np.float64); q = np.asarray(q, np.float64)
    return lib().pto_transformed_distance(_ptr(p), _ptr(q))


def min_distance_to_rectangle(p, lo, hi):
    p, lo, hi = (np.asarray(v, np.float64) for v in (p, lo, hi))
    d = np.zeros(3)
    return lib().pto_min_distance_to_rectangle(_ptr(p), _ptr(lo), _ptr(hi), _ptr(d)), d


def max_distance_to_rectangle(p, lo, hi):
    p, lo, hi = (np.asarray(v, np.float64) for v in (p, lo, hi))
    d = np.zeros(3)
    return lib().pto_max_distance_to_rectangle(_ptr(p), _ptr(lo), _ptr(hi), _ptr(d)), d


def new_distance(dist, old_off, new_off):
    return lib().pto_new_distance(dist, old_off, new_off)


# ---- k-NN --------------------------------------------------------------------
def knn_bruteforce(src, tgt, k, gidx=None, nthreads=0):
    src = _planar64(src); tgt = _planar64(tgt)
    n, m = src.shape[1], tgt.shape[1]
    idx = np.empty((m, k), np.uint32); d2 = np.empty((m, k), np.float64)
    g = None if gidx is None else np.ascontiguousarray(gidx, dtype=np.uint32)
    rc = lib().pto_knn_bruteforce(_ptr(src), n, _ptr(g), _ptr(tgt), m, k, _ptr(idx), _ptr(d2), nthreads)
    assert rc == 0
    return idx, d2


class KdTree:
    """CPU restatement of the reference's CGAL kd-tree search (timed CPU baseline)."""

    def __init__(self, src, gidx=None):
        self._src = _planar64(src)
        g = None if gidx is None else np.ascontiguousarray(gidx, dtype=np.uint32)
        self._h = lib().pto_kdtree_build(_ptr(self._src), self._src.shape[1], _ptr(g))

    def query(self, tgt, k, nthreads=0):
        tgt = _planar64(tgt)
        m = tgt.shape[1]
        idx = np.empty((m, k), np.uint32); d2 = np.empty((m, k), np.float64)
        rc = lib().pto_kdtree_query(self._h, _ptr(tgt), m, k, _ptr(idx), _ptr(d2), nthreads)
        assert rc == 0
        return idx, d2

    def close(self):
        if self._h:
            lib().pto_kdtree_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def num_threads():
    return lib().pto_num_threads()


def knn_numpy(src, tgt, k):
    """Third opinion in pure numpy (small sizes only): same (d2, idx) total order."""
    src = _planar64(src); tgt = _planar64(tgt)
    n, m = src.shape[1], tgt.shape[1]
    idx = np.full((m, k), NOIDX, np.uint32); d2 = np.full((m, k), np.inf)
    for t in range(m):
        dx = tgt[0, t] - src[0]; dy = tgt[1, t] - src[1]; dz = tgt[2, t] - src[2]
        d = (dx * dx + dy * dy) + dz * dz
        order = np.lexsort((np.arange(n), d))[:k]
        idx[t, :len(order)] = order
        d2[t, :len(order)] = d[order]
    return idx, d2


def merge_candidates(idx_lists, d2_lists):
    idx_lists = np.ascontiguousarray(idx_lists, np.uint32); d2_lists = np.ascontiguousarray(d2_lists, np.float64)
    g, m, k = idx_lists.shape
    idx = np.empty((m, k), np.uint32); d2 = np.empty((m, k), np.float64)
    lib().pto_merge_candidates(_ptr(idx_lists), _ptr(d2_lists), g, m, k, _ptr(idx), _ptr(d2))
    return idx, d2


# ---- blend / PCA (build-defined; see pt_oracle.c) ---------------------------
def blend(idx, d2, rgb, nrm, mode=0):
    idx = np.ascontiguousarray(idx, np.uint32); d2 = np.ascontiguousarray(d2, np.float64)
    m, k = idx.shape
    rgb = None if rgb is None else np.ascontiguousarray(rgb, np.uint8)
    nrm = None if nrm is None else np.ascontiguousarray(nrm, np.float32)
    ro = np.zeros((m, 3), np.float32); no = np.zeros((m, 3), np.float32)
    rc = lib().pto_blend(_ptr(idx), _ptr(d2), m, k, mode, _ptr(rgb), _ptr(nrm), _ptr(ro), _ptr(no))
    assert rc == 0
    return ro, no


def blend_weighted(idx, w, rgb, nrm):
    """The reference's mix formula (pointsTransfer.cpp:95-97) for k terms with caller-given weights."""
    idx = np.ascontiguousarray(idx, np.uint32); w = np.ascontiguousarray(w, np.float64)
    m, k = idx.shape
    rgb = None if rgb is None else np.ascontiguousarray(rgb, np.uint8)
    nrm = None if nrm is None else np.ascontiguousarray(nrm, np.float32)
    ro = np.zeros((m, 3), np.float32); no = np.zeros((m, 3), np.float32)
    rc = lib().pto_blend_weighted(_ptr(idx), _ptr(w), m, k, _ptr(rgb), _ptr(nrm), _ptr(ro), _ptr(no))
    assert rc == 0
    return ro, no


def pca_normals(idx, src, nrm=None):
    idx = np.ascontiguousarray(idx, np.uint32)
    src = _planar64(src)
    m, k = idx.shape
    nrm = None if nrm is None else np.ascontiguousarray(nrm, np.float32)
    out = np.zeros((m, 3), np.float32); plan = np.zeros(m)
    lib().pto_pca_normals(_ptr(idx), m, k, _ptr(src), src.shape[1], _ptr(nrm), _ptr(out), _ptr(plan))
    return out, plan


def bake_texture(src_xyz, src_rgb, vert_xyz, vert_uv, vert_rgb, faces, nbr_idx, resolution):
    """Per-face texture bake, the build's definition (pt_oracle.c: pto_bake_texture; reference pointsTransfer.cpp:462-581, :66-107).
    src_xyz (3, n) / vert_xyz (3, nv) planar, *_rgb (., 3) uint8, vert_uv (nv, 2), faces (nf, 3) int32, nbr_idx (nv, k) uint32.
    Returns the (R, R, 4) BGRA atlas."""
    sx = _planar64(src_xyz); vx = _planar64(vert_xyz)
    srgb = np.ascontiguousarray(src_rgb, np.uint8); vrgb = np.ascontiguousarray(vert_rgb, np.uint8)
    uv = np.ascontiguousarray(vert_uv, np.float64); fc = np.ascontiguousarray(faces, np.int32).reshape(-1, 3)
    nb = np.ascontiguousarray(nbr_idx, np.uint32)
    out = np.zeros((resolution, resolution, 4), np.uint8)
    rc = lib().pto_bake_texture(_ptr(sx), _ptr(srgb), sx.shape[1], _ptr(vx), _ptr(uv), _ptr(vrgb), vx.shape[1], _ptr(fc), fc.shape[0], _ptr(nb), nb.shape[1],
                                resolution, _ptr(out))
    assert rc == 0
    return out


def dilate_pad(bgra, ksize=25):
    """Edge padding (pointsTransfer.cpp:593-611): texture + (dilate(texture, ksize x ksize) & ~alpha), saturating."""
    a = np.ascontiguousarray(bgra, np.uint8)
    out = np.empty_like(a)
    assert lib().pto_dilate_pad(_ptr(a), a.shape[0], ksize, _ptr(out)) == 0
    return out


def ref_point_layout():
    """Layout facts printed by the probe built from the reference's own Point.h."""
    with open(os.path.join(_HERE, "_ref", "point_layout.json")) as f:
        return json.load(f)
