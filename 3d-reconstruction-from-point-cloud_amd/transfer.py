"""Host-side mirror of the reference's hot path, over the C ABI.

The reference has no operator/plugin interface for this path: it is three statements of main()
(reference src/pointsTransfer.cpp):
    :259      Tree tree(points.begin(), points.end());          -> PointsTransfer.build*()
    :474-478  K_neighbor_search search(tree, v, K); iterate      -> PointsTransfer.query*()
    :95-97    the only blend arithmetic (barycentric colour mix) -> PointsTransfer.blend()
Names below follow those steps.  Arrays are numpy on the host, or raw device pointers / torch
tensors (anything with .data_ptr()) for the *_dev methods.  Everything runs in libpt_hip.so on the
GPU; nothing here computes neighbours on the CPU.
"""
import ctypes as C
import math
import sys

import numpy as np

from . import capi

K_REFERENCE = 20   # `const unsigned int K = 20`, reference src/pointsTransfer.cpp:128


def _np_type(xyz_type):
    return {capi.F64: np.float64, capi.F16: np.float16}.get(xyz_type, np.float32)


def _planar(xyz, xyz_type=None):
    a = np.asarray(xyz)
    if a.ndim != 2 or a.shape[0] != 3:
        raise ValueError("xyz must be planar with shape (3, n)")
    if xyz_type is None:
        xyz_type = capi.F64 if a.dtype == np.float64 else (capi.F16 if a.dtype == np.float16 else capi.F32)
    return np.ascontiguousarray(a, dtype=_np_type(xyz_type)), xyz_type


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):      # torch tensor
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


class PointsTransfer:
    """One context = one GPU = one (slab of a) source cloud."""

    def __init__(self, device=0, rho=None, k_hint=None):
        self._L = capi.lib()
        self._h = C.c_void_p()
        dev = (C.c_int * 1)(device)
        rc = self._L.pt_ctx_create(C.byref(self._h), dev, 1)
        if rc != capi.OK:
            self._h = C.c_void_p()
            raise capi.PtError(rc, "pt_ctx_create failed (no usable gfx950 device? there is no CPU fallback)")
        if k_hint is not None:
            self.set_param("k_hint", k_hint)      # cell density suited to the k the queries will use
        if rho is not None:
            self.set_param("rho", rho)

    # -- plumbing ------------------------------------------------------------------------
    def _chk(self, rc):
        if rc != capi.OK:
            raise capi.PtError(rc, self._L.pt_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.pt_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_param(self, name, value):
        self._chk(self._L.pt_set_param(self._h, name.encode(), float(value)))

    def set_stream(self, hip_stream):
        self._chk(self._L.pt_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def _adopt_torch_stream(self):
        """Device-pointer entry points are fed by torch tensors: run on torch's CURRENT stream so that the kernels
        queue behind whatever produced those tensors (the context's own stream is non-blocking and would race)."""
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            self._chk(self._L.pt_set_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def synchronize(self):
        self._chk(self._L.pt_synchronize(self._h))

    def stats(self):
        s = capi.Stats()
        self._chk(self._L.pt_stats(self._h, C.byref(s)))
        return {f[0]: (list(getattr(s, f[0])) if f[0] in ("grid_dim", "ms_kernel") else getattr(s, f[0])) for f in s._fields_ if not f[0].startswith("_pad")}

    @property
    def num_source(self):
        return int(self._L.pt_num_source(self._h))

    @property
    def num_targets(self):
        return int(self._L.pt_num_targets(self._h))

    # -- build (pointsTransfer.cpp:259) -----------------------------------------------------
    def build_aos(self, points):
        """points: numpy structured/raw array of the reference's 80-byte Point records."""
        a = np.ascontiguousarray(points)
        assert a.dtype.itemsize == 80, "Point records are 80 bytes (reference src/Point.h)"
        self._chk(self._L.pt_build_aos(self._h, _ptr(a), a.shape[0]))

    def build(self, xyz, rgb=None, nrm=None, xyz_type=None, gidx=None):
        a, t = _planar(xyz, xyz_type)
        n = a.shape[1]
        if gidx is not None:
            g = np.ascontiguousarray(gidx, dtype=np.uint32)
            assert g.shape == (n,)
            self._chk(self._L.pt_build_soa_indexed(self._h, _ptr(a), t, _ptr(g), n, 0))
            return
        r = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.uint8)
        m = None if nrm is None else np.ascontiguousarray(nrm, dtype=np.float32)
        if r is not None:
            assert r.shape == (n, 3)
        if m is not None:
            assert m.shape == (n, 3)
        self._chk(self._L.pt_build_soa(self._h, _ptr(a), t, _ptr(r), _ptr(m), n, 0))

    def set_attributes(self, rgb, nrm):
        r = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.uint8)
        m = None if nrm is None else np.ascontiguousarray(nrm, dtype=np.float32)
        n = (r if r is not None else m).shape[0]
        self._chk(self._L.pt_set_attributes(self._h, _ptr(r), _ptr(m), n, 0))

    def set_attributes_local(self, rgb, nrm):
        """The attribute records of THIS slab's points only, in the order of the slab's arrays (after set_param("local_ids", 1) and a
        build with strictly ascending global indices): 16 n bytes on the GPU instead of 16 N."""
        r = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.uint8)
        m = None if nrm is None else np.ascontiguousarray(nrm, dtype=np.float32)
        self._chk(self._L.pt_set_attributes_local(self._h, _ptr(r), _ptr(m), 0))

    def set_attributes_range(self, first, rgb, nrm, n_total):
        """Records [first, first + len) of the attribute table of n_total points, from host arrays (rgb [c,3] u8, nrm [c,3] f32)."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8) if rgb is not None else None
        nrm = np.ascontiguousarray(nrm, dtype=np.float32) if nrm is not None else None
        count = len(rgb) if rgb is not None else len(nrm)
        self._chk(self._L.pt_set_attributes_range(self._h, int(first), int(count), _ptr(rgb), _ptr(nrm), int(n_total)))

    def build_synth(self, n_total, seed, xyz_type=capi.F32, dist=capi.DIST_UNIFORM, slab_axis=-1, slab_lo=-math.inf, slab_hi=math.inf):
        self._chk(self._L.pt_build_synth(self._h, n_total, seed, dist, xyz_type, slab_axis, slab_lo, slab_hi))

    def rebuild(self):
        self._chk(self._L.pt_rebuild(self._h))

    # -- query (pointsTransfer.cpp:462-479) ---------------------------------------------------
    def query(self, targets, k=K_REFERENCE, xyz_type=None, want_d2=True):
        a, t = _planar(targets, xyz_type)
        m = a.shape[1]
        idx = np.empty((m, k), np.uint32)
        d2 = np.empty((m, k), np.float64) if want_d2 else None
        self._chk(self._L.pt_query_soa(self._h, _ptr(a), t, m, k, 0, _ptr(idx), _ptr(d2)))
        return (idx, d2) if want_d2 else idx

    def query_aos(self, points, k=K_REFERENCE):
        a = np.ascontiguousarray(points)
        assert a.dtype.itemsize == 80
        m = a.shape[0]
        idx = np.empty((m, k), np.uint32)
        d2 = np.empty((m, k), np.float64)
        self._chk(self._L.pt_query_aos(self._h, _ptr(a), m, k, _ptr(idx), _ptr(d2)))
        return idx, d2

    def targets_synth(self, m_total, seed, xyz_type=capi.F32, dist=capi.DIST_UNIFORM, slab_axis=-1, slab_lo=-math.inf, slab_hi=math.inf):
        self._chk(self._L.pt_targets_synth(self._h, m_total, seed, dist, xyz_type, slab_axis, slab_lo, slab_hi))

    def set_targets(self, xyz, xyz_type=None):
        """Make the caller's targets resident: planar (3, m) numpy array (host) or torch tensor (device, pass xyz_type)."""
        if hasattr(xyz, "data_ptr"):
            assert xyz_type is not None and xyz.dim() == 2 and xyz.shape[0] == 3 and xyz.is_contiguous()
            self._adopt_torch_stream()
            self._chk(self._L.pt_targets_soa(self._h, _ptr(xyz), xyz_type, xyz.shape[1], 1))
            return
        a, t = _planar(xyz, xyz_type)
        self._chk(self._L.pt_targets_soa(self._h, _ptr(a), t, a.shape[1], 0))

    def set_targets_aos(self, points):
        a = np.ascontiguousarray(points)
        assert a.dtype.itemsize == 80
        self._chk(self._L.pt_targets_aos(self._h, _ptr(a), a.shape[0]))

    def query_resident_dev(self, k, idx_dev, d2_dev=None):
        self._adopt_torch_stream()
        self._chk(self._L.pt_query_resident(self._h, k, _ptr(idx_dev), _ptr(d2_dev)))

    def query_blend_resident_dev(self, k, mode, idx_dev, d2_dev, rgb_out_dev, nrm_out_dev):
        """query_resident_dev + blend_dev in one pass (the tile kernel gathers the attributes as it settles a target)."""
        self._adopt_torch_stream()
        self._chk(self._L.pt_query_blend_resident(self._h, k, mode, _ptr(idx_dev), _ptr(d2_dev), _ptr(rgb_out_dev), _ptr(nrm_out_dev)))

    def query_dev(self, xyz_dev, xyz_type, m, k, idx_dev, d2_dev=None):
        self._adopt_torch_stream()
        self._chk(self._L.pt_query_soa(self._h, _ptr(xyz_dev), xyz_type, m, k, 1, _ptr(idx_dev), _ptr(d2_dev)))

    def query_bounded_dev(self, xyz_dev, xyz_type, bound2_dev, m, k, idx_dev, d2_dev):
        self._adopt_torch_stream()
        self._chk(self._L.pt_query_bounded_dev(self._h, _ptr(xyz_dev), xyz_type, _ptr(bound2_dev), m, k, _ptr(idx_dev), _ptr(d2_dev)))

    def resident_target_ids_dev(self, ids_dev):
        self._adopt_torch_stream()
        self._chk(self._L.pt_resident_target_ids(self._h, _ptr(ids_dev)))

    def resident_target_xyz_dev(self, xyz_dev):
        self._adopt_torch_stream()
        self._chk(self._L.pt_resident_target_xyz(self._h, _ptr(xyz_dev)))

    def resident_source_xyz_dev(self, xyz_dev):
        """Planar xyz of the resident source cloud as kept on the device (fp16 clouds stay fp16); returns the element type."""
        self._adopt_torch_stream()
        t = C.c_int(-1)
        self._chk(self._L.pt_resident_source_xyz(self._h, _ptr(xyz_dev), C.byref(t)))
        return int(t.value)

    # -- blend / PCA ----------------------------------------------------------------------------
    def blend(self, idx, d2=None, mode=capi.BLEND_MEAN):
        idx = np.ascontiguousarray(idx, np.uint32)
        m, k = idx.shape
        d2c = None if d2 is None else np.ascontiguousarray(d2, np.float64)
        rgb = np.empty((m, 3), np.float32)
        nrm = np.empty((m, 3), np.float32)
        self._chk(self._L.pt_blend(self._h, _ptr(idx), _ptr(d2c), m, k, mode, _ptr(rgb), _ptr(nrm)))
        return rgb, nrm

    def blend_weighted(self, idx, w):
        """out = sum_j w[t, j] * a[idx[t, j]] -- the reference's mix formula (src/pointsTransfer.cpp:95-97) with caller weights."""
        idx = np.ascontiguousarray(idx, np.uint32); w = np.ascontiguousarray(w, np.float64)
        m, k = idx.shape
        assert w.shape == (m, k)
        rgb = np.empty((m, 3), np.float32); nrm = np.empty((m, 3), np.float32)
        self._chk(self._L.pt_blend_weighted(self._h, _ptr(idx), _ptr(w), m, k, _ptr(rgb), _ptr(nrm)))
        return rgb, nrm

    def blend_dev(self, idx_dev, d2_dev, m, k, mode, rgb_out_dev, nrm_out_dev):
        self._adopt_torch_stream()
        self._chk(self._L.pt_blend_dev(self._h, _ptr(idx_dev), _ptr(d2_dev), m, k, mode, _ptr(rgb_out_dev), _ptr(nrm_out_dev)))

    def pca_normals(self, idx):
        idx = np.ascontiguousarray(idx, np.uint32)
        m, k = idx.shape
        out = np.empty((m, 3), np.float32)
        self._chk(self._L.pt_pca_normals(self._h, _ptr(idx), m, k, _ptr(out)))
        return out

    def pca_normals_dev(self, idx_dev, m, k, nrm_out_dev):
        self._adopt_torch_stream()
        self._chk(self._L.pt_pca_normals_dev(self._h, _ptr(idx_dev), m, k, _ptr(nrm_out_dev)))

    # -- native slab exchange over RCCL ---------------------------------------------
    @staticmethod
    def comm_unique_id():
        """128 bytes every rank needs for comm_init (create on one rank, share out of band -- e.g. a torch.distributed broadcast)."""
        buf = C.create_string_buffer(128)
        rc = capi.lib().pt_comm_unique_id(buf)
        if rc != capi.OK:
            raise capi.PtError(rc, "pt_comm_unique_id failed (librccl not loadable?)")
        return bytes(buf.raw)

    def comm_init(self, world, rank, uid):
        assert len(uid) == 128
        self._chk(self._L.pt_comm_init(self._h, world, rank, C.create_string_buffer(bytes(uid), 128)))

    def comm_destroy(self):
        self._chk(self._L.pt_comm_destroy(self._h))

    def comm_abort(self):
        """Error-path teardown (ncclCommAbort): never waits for peers."""
        self._chk(self._L.pt_comm_abort(self._h))

    def exchange_merge_dev(self, xyz_dev, xyz_type, m, k, axis, bounds, idx_dev, d2_dev, blend_mode=-1, rgb_dev=None, nrm_dev=None, want_stats=True):
        """Complete the home-slab lists idx_dev / d2_dev ([m, k], in place) with the other ranks' candidates over RCCL; returns
        the exchange counters.  blend_mode >= 0 redoes the blend of the completed rows into rgb_dev / nrm_dev.
        want_stats=False passes no stats struct: the call then only enqueues after its one count read-back (no trailing host wait)
        and returns {}."""
        self._adopt_torch_stream()
        b = (C.c_double * len(bounds))(*bounds)
        st = capi.ExchangeStats() if want_stats else None
        self._chk(self._L.pt_exchange_merge_dev(self._h, _ptr(xyz_dev), xyz_type, m, k, axis, b, _ptr(idx_dev), _ptr(d2_dev), blend_mode,
                                                _ptr(rgb_dev), _ptr(nrm_dev), C.byref(st) if want_stats else None))
        return {f[0]: getattr(st, f[0]) for f in st._fields_} if want_stats else {}

    @staticmethod
    def exchange_merge_local(pts, xyz_devs, xyz_type, k, axis, bounds, idx_devs, d2_devs, blend_mode=-1, rgb_devs=None, nrm_devs=None):
        """The same protocol for G contexts of this process (G logical slabs on one GPU), device copies as the transport."""
        g = len(pts)
        for p_ in pts:
            p_._adopt_torch_stream()
        arr = lambda items: (C.c_void_p * g)(*[_ptr(t) for t in items])
        ms = (C.c_uint64 * g)(*[int(x.shape[1]) for x in xyz_devs])
        b = (C.c_double * len(bounds))(*bounds)
        rc = capi.lib().pt_exchange_merge_local((C.c_void_p * g)(*[p_._h for p_ in pts]), g, arr(xyz_devs), xyz_type, ms, k, axis, b, arr(idx_devs), arr(d2_devs),
                                                blend_mode, arr(rgb_devs) if rgb_devs else None, arr(nrm_devs) if nrm_devs else None)
        if rc != capi.OK:
            raise capi.PtError(rc, "; ".join(capi.lib().pt_last_error(p_._h).decode() for p_ in pts))

    # -- out-of-core source (README.md:3 "billions of points") ---------------------------------------
    def stream_query(self, xyz, chunk_points, k=K_REFERENCE, first_id=0, xyz_type=None):
        """k-NN of the RESIDENT targets in a cloud kept in host memory and streamed through the GPU chunk by chunk;
        returns (idx uint64 (m, k), d2 (m, k)) -- indices are first_id + position in `xyz`."""
        a, t = _planar(xyz, xyz_type)
        m = self.num_targets
        idx = np.empty((m, k), np.uint64); d2 = np.empty((m, k), np.float64)
        self._chk(self._L.pt_stream_query(self._h, _ptr(a), t, a.shape[1], int(chunk_points), int(first_id), k, _ptr(idx), _ptr(d2)))
        return idx, d2

    # -- texture bake (pointsTransfer.cpp:466-615) ------------------------------------------------
    def bake_texture(self, mesh_vertices, faces, nbr_idx, resolution=8192, pad_ksize=0):
        """mesh_vertices: POINT_DTYPE records (ver, color, U, V are read); faces: int32 (F, 3); nbr_idx: uint32 (V, k) from a
        query of those vertices.  Returns the (resolution, resolution, 4) BGRA atlas; pad_ksize > 0 applies the edge padding."""
        v = np.ascontiguousarray(mesh_vertices)
        assert v.dtype.itemsize == 80
        f = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
        nb = np.ascontiguousarray(nbr_idx, dtype=np.uint32)
        assert nb.ndim == 2 and nb.shape[0] == v.shape[0]
        out = np.empty((resolution, resolution, 4), np.uint8)
        self._chk(self._L.pt_bake_texture(self._h, _ptr(v), v.shape[0], _ptr(f), f.shape[0], _ptr(nb), nb.shape[1], resolution, pad_ksize, _ptr(out)))
        return out

    def texture_pad(self, bgra, ksize=25):
        a = np.ascontiguousarray(bgra, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[0] == a.shape[1] and a.shape[2] == 4
        out = np.empty_like(a)
        self._chk(self._L.pt_texture_pad(self._h, _ptr(a), a.shape[0], ksize, _ptr(out)))
        return out

    # -- multi-GPU helpers ---------------------------------------------------------
    def merge_candidates_dev(self, idx_lists_dev, d2_lists_dev, g, m, k, idx_out_dev, d2_out_dev):
        self._adopt_torch_stream()
        self._chk(self._L.pt_merge_candidates_dev(self._h, _ptr(idx_lists_dev), _ptr(d2_lists_dev), g, m, k, _ptr(idx_out_dev), _ptr(d2_out_dev)))

    def pack_requests_dev(self, tgt_xyz_dev, xyz_type, d2_dev, m, k, slab_axis, slab_bounds, my_slab, sel_dev, pkt_dev):
        """slab_need + selection in one pass: returns the number of request packets written to pkt_dev[:c] / sel_dev[:c]."""
        self._adopt_torch_stream()
        b = np.ascontiguousarray(slab_bounds, np.float64)
        cnt = C.c_uint32(0)
        self._chk(self._L.pt_pack_requests_dev(self._h, _ptr(tgt_xyz_dev), xyz_type, _ptr(d2_dev), m, k, slab_axis, _ptr(b), b.shape[0] - 1, my_slab,
                                               _ptr(sel_dev), _ptr(pkt_dev), C.byref(cnt)))
        return int(cnt.value)

    def slab_need_dev(self, tgt_xyz_dev, xyz_type, d2_dev, m, k, slab_axis, slab_bounds, my_slab, need_dev):
        self._adopt_torch_stream()
        b = np.ascontiguousarray(slab_bounds, np.float64)
        g = b.shape[0] - 1
        self._chk(self._L.pt_slab_need_dev(self._h, _ptr(tgt_xyz_dev), xyz_type, _ptr(d2_dev), m, k, slab_axis, _ptr(b), g, my_slab, _ptr(need_dev)))


POINT_DTYPE = np.dtype({
    "names": ["ver", "normal", "color", "U", "V"],
    "formats": [(np.float64, 3), (np.float64, 3), (np.int32, 3), np.float64, np.float64],
    "offsets": [0, 24, 48, 64, 72],
    "itemsize": 80,
})
"""numpy view of the reference's Point record (reference src/Point.h:2-6)."""
