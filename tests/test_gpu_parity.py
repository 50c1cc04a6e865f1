"""GPU suite: the HIP path, called through the C ABI, against the CPU oracle and the golden fixtures.
Bar: neighbour indices and squared distances bit-exact; blended attributes within 1e-5 (colour/255, normals)."""
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _check_exact(got, want, what=""):
    gi, gd = got
    wi, wd = want
    assert np.array_equal(gi, wi), "%s: indices differ in %d of %d rows" % (what, (gi != wi).any(axis=1).sum(), gi.shape[0])
    assert np.array_equal(gd, wd), "%s: d2 differ" % what


@pytest.fixture(scope="module")
def pt(pkg):
    p = pkg.PointsTransfer(device=0)
    yield p
    p.close()


# ---- golden fixtures and the reference-shaped cases -----------------------------------------------------
@pytest.mark.parametrize("name,ks", [("c1", (1, 8, 16, 32)), ("ties", (1, 8, 20)), ("outside", (8,)), ("tiny", (8, 32)), ("flat", (8,))])
def test_golden_cases(pt, golden, golden_cases, name, ks):
    src, tgt = golden_cases[name]
    pt.build(src)
    for k in ks:
        sub = tgt[:, :256] if (name == "c1" and k > 8) else tgt
        idx, d2 = pt.query(sub, k)
        _check_exact((idx, d2), (golden["%s_k%d_idx" % (name, k)], golden["%s_k%d_d2" % (name, k)]), "%s k=%d" % (name, k))


@pytest.mark.parametrize("n,m,k,seed", [(10000, 1000, 1, 0xC1), (200000, 20000, 8, 0xC2), (300000, 5000, 16, 0xC3), (150000, 3000, 32, 0xC5), (70000, 4000, 20, 7)])
def test_seeded_uniform_vs_oracle(pt, oracle, n, m, k, seed):
    src, tgt = oracle.synth_xyz(seed, 0, n), oracle.synth_xyz(seed, 1, m)
    pt.build(src)
    got = pt.query(tgt, k)
    kd = oracle.KdTree(src)
    _check_exact(got, kd.query(tgt, k), "uniform n=%d k=%d" % (n, k))
    if n <= 70000:
        _check_exact(got, oracle.knn_bruteforce(src, tgt, k), "brute force")


@pytest.mark.parametrize("rho", [1.0, 3.0, 20.0, 200.0])
def test_cell_size_does_not_change_results(pkg, oracle, rho):
    src, tgt = oracle.synth_xyz(11, 0, 60000), oracle.synth_xyz(11, 1, 3000)
    with pkg.PointsTransfer(device=0, rho=rho) as p:
        p.build(src)
        _check_exact(p.query(tgt, 8), oracle.knn_bruteforce(src, tgt, 8), "rho=%g" % rho)


@pytest.mark.parametrize("k", [1, 8, 13, 16, 20, 32])
def test_tile_kernel_and_group_kernel_agree(pkg, oracle, k):
    """The two k-NN kernels (a DPP quad per target over an LDS-staged tile / 8 lanes per target) must give the same
    bits as each other and as the oracle; the tile kernel may only hand a small share of the targets over."""
    src, tgt = oracle.synth_xyz(21, 0, 400000), oracle.synth_xyz(21, 1, 30000)
    want = oracle.KdTree(src).query(tgt, k)
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build(src)
        p.set_param("tile", 1)
        got_tile = p.query(tgt, k)
        left = p.stats()["n_leftover"]
        p.set_param("tile", 0)
        got_group = p.query(tgt, k)
        assert p.stats()["n_leftover"] == 0
    _check_exact(got_tile, want, "tile k=%d" % k)
    _check_exact(got_group, want, "group k=%d" % k)
    # k in 25..32 runs at a density where ring 1 is only just enough: up to a quarter may go to the group kernel
    assert left < (0.05 if k <= 24 else 0.25) * tgt.shape[1], "tile kernel handed over %d of %d targets" % (left, tgt.shape[1])


def test_device_generator_matches_oracle(pkg, oracle):
    import torch
    n, m, k, seed = 100000, 10000, 8, 0xC2
    with pkg.PointsTransfer(device=0) as p:
        p.build_synth(n, seed)
        p.targets_synth(m, seed)
        assert p.num_source == n and p.num_targets == m
        xyz = torch.empty((3, m), dtype=torch.float32, device="cuda")
        p.resident_target_xyz_dev(xyz)
        assert np.array_equal(xyz.cpu().numpy(), oracle.synth_xyz(seed, 1, m))       # bit-identical generator
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda")
        d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        src, tgt = oracle.synth_xyz(seed, 0, n), oracle.synth_xyz(seed, 1, m)
        want = oracle.KdTree(src).query(tgt, k)
        _check_exact((idx.cpu().numpy().view(np.uint32), d2.cpu().numpy()), want, "device generator")
        # attributes generated on the device: colours exact, normals to rounding
        rgb = torch.empty((m, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((m, 3), dtype=torch.float32, device="cuda")
        p.blend_dev(idx, d2, m, k, pkg.BLEND_MEAN, rgb, nrm)
        torch.cuda.synchronize()
        rc, rn = oracle.blend(want[0], want[1], oracle.synth_rgb(seed, n), oracle.synth_nrm(seed, n), 0)
        assert np.abs(rgb.cpu().numpy() - rc).max() / 255 <= TOL and np.abs(nrm.cpu().numpy() - rn).max() <= TOL


def test_transient_queries_leave_resident_targets_alone(pkg, oracle):
    import torch
    n, m, k, seed = 50000, 4000, 8, 5
    with pkg.PointsTransfer(device=0) as p:
        p.build_synth(n, seed)
        p.targets_synth(m, seed)
        a_i = torch.empty((m, k), dtype=torch.int32, device="cuda"); a_d = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, a_i, a_d)
        other = oracle.synth_xyz(99, 1, 777)
        p.query(other, k)                                              # host targets
        ox = torch.from_numpy(other).cuda()
        bi = torch.empty((777, k), dtype=torch.int32, device="cuda"); bd = torch.empty((777, k), dtype=torch.float64, device="cuda")
        p.query_bounded_dev(ox, pkg.F32, torch.full((777,), 1e-3, dtype=torch.float64, device="cuda"), 777, k, bi, bd)
        assert p.num_targets == m
        b_i = torch.empty_like(a_i); b_d = torch.empty_like(a_d)
        p.query_resident_dev(k, b_i, b_d)
        torch.cuda.synchronize()
        assert torch.equal(a_i, b_i) and torch.equal(a_d, b_d)
        # the bounded query returns exactly the neighbours within the bound
        wi, wd = oracle.knn_bruteforce(oracle.synth_xyz(seed, 0, n), other, k)
        out = wd > 1e-3
        wi[out] = 0xFFFFFFFF; wd[out] = np.inf
        assert np.array_equal(bi.cpu().numpy().view(np.uint32), wi) and np.array_equal(bd.cpu().numpy(), wd)


# ---- edge cases the domain has ---------------------------------------------------------------------------
def test_duplicates_and_exact_ties(pt, oracle):
    rng = np.random.default_rng(1)
    base = rng.integers(0, 4, size=(3, 500)).astype(np.float32) / 4          # 64 distinct positions, ~8 copies each
    src = np.concatenate([base, base, base], axis=1)                         # every point three times
    tgt = rng.integers(0, 8, size=(3, 300)).astype(np.float32) / 8
    pt.build(src)
    for k in (1, 5, 8, 20, 32):
        _check_exact(pt.query(tgt, k), oracle.knn_bruteforce(src, tgt, k), "dups k=%d" % k)


def test_all_points_identical_and_single_point(pt, oracle):
    src = np.full((3, 100), 0.5, np.float32)
    tgt = np.array([[0.5, 0.1, 3.0], [0.5, 0.2, -2.0], [0.5, 0.3, 0.5]], np.float32)
    pt.build(src)
    _check_exact(pt.query(tgt, 8), oracle.knn_bruteforce(src, tgt, 8), "identical")
    src1 = np.array([[0.25], [0.5], [0.75]], np.float32)
    pt.build(src1)
    _check_exact(pt.query(tgt, 4), oracle.knn_bruteforce(src1, tgt, 4), "single")


def test_empty_inputs(pt):
    pt.build(np.zeros((3, 0), np.float32))
    idx, d2 = pt.query(np.zeros((3, 5), np.float32) + 0.5, 4)
    assert (idx == 0xFFFFFFFF).all() and np.isinf(d2).all()
    pt.build(np.random.default_rng(0).random((3, 100)).astype(np.float32))
    idx, d2 = pt.query(np.zeros((3, 0), np.float32), 4)
    assert idx.shape == (0, 4) and d2.shape == (0, 4)


def test_ragged_density_and_far_targets(pt, oracle):
    rng = np.random.default_rng(9)
    blob = (0.5 + 0.01 * rng.standard_normal((3, 30000))).astype(np.float32)       # one dense blob ...
    sparse = rng.random((3, 2000)).astype(np.float32)                               # ... in a sparse box
    src = np.concatenate([blob, sparse], axis=1)
    tgt = np.concatenate([rng.random((3, 500)), 0.5 + 0.02 * rng.standard_normal((3, 500)), 3 * rng.random((3, 100)) - 1], axis=1).astype(np.float32)
    pt.build(src)
    for k in (8, 20):
        _check_exact(pt.query(tgt, k), oracle.KdTree(src).query(tgt, k), "ragged k=%d" % k)


def test_k_bounds_and_call_order(pkg, pt):
    with pkg.PointsTransfer(device=0) as p:
        with pytest.raises(pkg.PtError) as e:
            p.query(np.zeros((3, 1), np.float32), 4)
        assert e.value.code == pkg.capi.ERR_STATE
    pt.build(np.random.default_rng(0).random((3, 50)).astype(np.float32))
    for bad in (0, 33):
        with pytest.raises(pkg.PtError) as e:
            pt.query(np.zeros((3, 1), np.float32), bad)
        assert e.value.code == pkg.capi.ERR_ARG
    with pytest.raises(pkg.PtError):
        pt.query(np.zeros((3, 1), np.float64), 4)          # f64 targets against an f32 cloud would need narrowing: refused


# ---- the reference's own record type: AoS Point, double coordinates ------------------------------------------
def test_aos_point_records_double_path(pkg, pt, oracle):
    rng = np.random.default_rng(4)
    n, m, k = 40000, 2000, 20                                                    # K = 20 as the reference
    cloud = np.zeros(n, dtype=pkg.POINT_DTYPE)
    cloud["ver"] = rng.random((n, 3)) * [10.0, 3.0, 1.0] - [5.0, 0, 0]           # doubles that are not fp32-representable
    cloud["normal"] = rng.standard_normal((n, 3))
    cloud["color"] = rng.integers(0, 256, (n, 3))
    verts = np.zeros(m, dtype=pkg.POINT_DTYPE)
    verts["ver"] = rng.random((m, 3)) * [10.0, 3.0, 1.0] - [5.0, 0, 0]
    pt.build_aos(cloud)
    idx, d2 = pt.query_aos(verts, k)
    want = oracle.knn_bruteforce(cloud["ver"].T, verts["ver"].T, k)
    _check_exact((idx, d2), want, "AoS f64")
    c, nn = pt.blend(idx, d2, pkg.BLEND_MEAN)
    rc, rn = oracle.blend(want[0], want[1], cloud["color"].astype(np.uint8), cloud["normal"].astype(np.float32), 0)
    assert np.abs(c - rc).max() / 255 <= TOL and np.abs(nn - rn).max() <= TOL
    # planar f64 entry point gives the same
    pt.build(cloud["ver"].T.copy(), xyz_type=pkg.F64)
    _check_exact(pt.query(verts["ver"].T.copy(), k, xyz_type=pkg.F64), want, "planar f64")
    # fp32 targets against the double cloud are widened exactly
    t32 = verts["ver"].T.astype(np.float32)
    _check_exact(pt.query(t32, k), oracle.knn_bruteforce(cloud["ver"].T, t32.astype(np.float64), k), "f32 targets on f64 cloud")


def test_fp16_coordinates(pkg, pt, oracle):
    """BASELINE config 5's coordinate type: fp16 xyz, ranked exactly like the oracle on the widened values."""
    import torch
    rng = np.random.default_rng(16)
    src = rng.random((3, 60000)).astype(np.float16)          # heavy duplication: only ~1000 distinct values per axis
    tgt = rng.random((3, 3000)).astype(np.float16)
    pt.build(src)
    got = pt.query(tgt, 8)
    _check_exact(got, oracle.knn_bruteforce(src.astype(np.float64), tgt.astype(np.float64), 8), "fp16 host arrays")
    with pkg.PointsTransfer(device=0) as p:                    # the on-device generator rounds to fp16 the same way
        n, m, k, seed = 80000, 5000, 8, 0xC5
        p.build_synth(n, seed, xyz_type=pkg.F16)
        p.targets_synth(m, seed, xyz_type=pkg.F16)
        xyz = torch.empty((3, m), dtype=torch.float32, device="cuda")
        p.resident_target_xyz_dev(xyz)
        t16 = oracle.synth_xyz(seed, 1, m).astype(np.float16)
        assert np.array_equal(xyz.cpu().numpy(), t16.astype(np.float32))
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        s16 = oracle.synth_xyz(seed, 0, n).astype(np.float16)
        _check_exact((idx.cpu().numpy().view(np.uint32), d2.cpu().numpy()), oracle.KdTree(s16.astype(np.float64)).query(t16.astype(np.float64), k), "fp16 generator")


def test_cli_end_to_end(tmp_path, pkg, oracle):
    """The C++ pointsTransfer CLI (reference argv + stdout lines) over the C ABI, on ASCII PLY files of the reference's
    grammar (cloud: x y z nx ny nz r g b; mesh: x y z nx ny nz u v r g b + faces), K = 20 as the reference."""
    import os, subprocess
    rng = np.random.default_rng(12)
    n, m, k = 5000, 300, 20
    cloud = np.round(rng.random((n, 3)) * [4, 2, 1], 6)                    # decimal text -> doubles that are not fp32 values
    cnrm = np.round(rng.standard_normal((n, 3)), 6); crgb = rng.integers(0, 256, (n, 3))
    verts = np.round(rng.random((m, 3)) * [4, 2, 1], 6)
    pc, mesh = tmp_path / "cloud.ply", tmp_path / "mesh.ply"
    with open(pc, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\n"
                "property float ny\nproperty float nz\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n)
        for p_, q_, c_ in zip(cloud, cnrm, crgb):
            f.write("%.6f %.6f %.6f %.6f %.6f %.6f %d %d %d\n" % (*p_, *q_, *c_))
    with open(mesh, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\n"
                "property float ny\nproperty float nz\nproperty float s\nproperty float t\nproperty uchar red\nproperty uchar green\n"
                "property uchar blue\nelement face 2\nproperty list uchar int vertex_indices\nend_header\n" % m)
        for v in verts:
            f.write("%.6f %.6f %.6f 0 0 1 0.5 0.5 1 2 3\n" % tuple(v))
        f.write("3 0 1 2\n3 2 1 3\n")
    exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "pointsTransfer")
    nb = tmp_path / "nn.bin"
    r = subprocess.run([exe, str(pc), str(mesh), "--neighbors", str(nb), "--out", str(tmp_path / "out.ply"), "--resolution", "256", "--json", str(tmp_path / "run.json")],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    import json
    rep = json.load(open(tmp_path / "run.json"))                            # --json: the stdout lines' seconds + pt_stats
    assert rep["k"] == k and rep["points"] == n and rep["mesh_vertices"] == m and rep["pt_stats"]["n_source"] == n
    assert set(rep["seconds"]) == {"read_cloud", "build", "read_mesh", "search", "blend", "bake", "output", "total"} and len(rep["pt_stats"]["ms_kernel"]) == 8
    assert os.path.getsize(tmp_path / "texture.png") > 100                 # the reference's artefact, in the working directory (:615)
    lines = [l.split(":")[0] for l in r.stdout.strip().splitlines()]
    assert lines == ["PC Point count", "Read point set in", "Built Kd tree in", "Mesh vertex count", "Mesh face count", "Read mesh faces",
                     "Neighbor search total time", "Draw triangles total time", "Output time", "Total real time", "VIRT", "RES"]   # reference order
    assert "PC Point count: %d" % n in r.stdout and "Mesh vertex count: %d" % m in r.stdout and "Mesh face count: 2" in r.stdout
    got = np.fromfile(nb, dtype=np.uint32).reshape(m, k)
    want, wd = oracle.knn_bruteforce(cloud.T, verts.T, k)                 # the doubles atof/strtod produce from the text
    assert np.array_equal(got, want)
    out = [l.split() for l in open(tmp_path / "out.ply").read().split("end_header\n")[1].strip().splitlines()]
    rc, rn = oracle.blend(want, wd, crgb.astype(np.uint8), cnrm.astype(np.float32), 0)
    col = np.array([[int(v) for v in row[8:11]] for row in out[:m]])
    assert np.abs(col - np.floor(rc)).max() <= 1                           # float -> uchar truncation, as the reference's rasteriser


@pytest.mark.parametrize("xyz_type,k", [("f32", 8), ("f16", 32)])
def test_clustered_distribution(pkg, oracle, xyz_type, k):
    """BASELINE config 5's shape at test size: clustered density (thin patches + blobs + a little uniform), targets a
    jittered subsample of the sources, fp16 coordinates with k = 32.  The device generator must reproduce the oracle's bits,
    and the (slow on this distribution, but exact) search must match the oracle."""
    import torch
    n, m, seed = 300000, 15000, 0xC5
    t = pkg.F16 if xyz_type == "f16" else pkg.F32
    src = oracle.synth_xyz(seed, 0, n, dist=1)
    tgt = oracle.synth_xyz(seed, 1, m, dist=1, n_total=n, m_total=m)
    if xyz_type == "f16":
        src = src.astype(np.float16).astype(np.float32); tgt = tgt.astype(np.float16).astype(np.float32)
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build_synth(n, seed, xyz_type=t, dist=pkg.capi.DIST_CLUSTERED)
        p.targets_synth(m, seed, xyz_type=t, dist=pkg.capi.DIST_CLUSTERED)
        xyz = torch.empty((3, m), dtype=torch.float32, device="cuda")
        p.resident_target_xyz_dev(xyz)
        assert np.array_equal(xyz.cpu().numpy(), tgt), "clustered target generator differs from the oracle's"
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
    want = oracle.KdTree(src).query(tgt, k)
    _check_exact((idx.cpu().numpy().view(np.uint32), d2.cpu().numpy()), want, "clustered %s k=%d" % (xyz_type, k))
    assert 0.0 <= src.min() and src.max() <= 1.0          # (fp16 rounding can reach 1.0 exactly)


def test_adaptive_cell_refinement(pkg, oracle):
    """A surface-like cloud (what real scans are) fills few cells of a volume-sized grid: the build must notice and refine;
    a uniform cloud must not be touched.  Results stay exact either way."""
    rng = np.random.default_rng(33)
    n = 400000
    src = rng.random((3, n)).astype(np.float32)
    src[2] = (0.5 + 0.2 * np.sin(3 * src[0]) * np.cos(2 * src[1]) + 1e-4 * rng.standard_normal(n)).astype(np.float32)   # a thin sheet
    tgt = src[:, ::40] + (1e-3 * rng.standard_normal((3, n // 40))).astype(np.float32)
    want = oracle.KdTree(src).query(tgt, 8)
    for adaptive in (1, 0):
        with pkg.PointsTransfer(device=0) as p:
            p.set_param("adaptive", adaptive)
            p.build(src)
            st = p.stats()
            _check_exact(p.query(tgt, 8), want, "sheet adaptive=%d" % adaptive)
        if adaptive:
            assert st["n_refine"] >= 1 and st["rho_occupied"] < 40, st
        else:
            assert st["n_refine"] == 0
    with pkg.PointsTransfer(device=0) as p:
        p.build(oracle.synth_xyz(3, 0, 300000))
        st = p.stats()
        assert st["n_refine"] == 0 and 3.5 < st["rho_occupied"] < 5.5, st


# ---- blend and PCA -------------------------------------------------------------------------------------------
def test_blend_modes_match_golden(pt, oracle, golden, golden_cases):
    src, tgt = golden_cases["c1"]
    rgb, nrm = oracle.synth_rgb(0xC1, 10000), oracle.synth_nrm(0xC1, 10000)
    pt.build(src, rgb, nrm)
    idx, d2 = pt.query(tgt, 8)
    for mode in (0, 1):
        c, n = pt.blend(idx, d2, mode)
        assert np.abs(c - golden["c1_k8_blend%d_rgb" % mode]).max() / 255 <= TOL
        assert np.abs(n - golden["c1_k8_blend%d_nrm" % mode]).max() <= TOL
    # missing neighbours (k > N) are skipped by the blend
    pt.build(src[:, :5], rgb[:5], nrm[:5])
    idx, d2 = pt.query(tgt[:, :10], 8)
    c, n = pt.blend(idx, d2, 0)
    rc, rn = oracle.blend(idx, d2, rgb[:5], nrm[:5], 0)
    assert np.abs(c - rc).max() / 255 <= TOL and np.abs(n - rn).max() <= TOL


def test_pca_matches_golden(pt, oracle, golden, golden_cases):
    src, tgt = golden_cases["c1"]
    pt.build(src, oracle.synth_rgb(0xC1, 10000), oracle.synth_nrm(0xC1, 10000))
    got = pt.pca_normals(golden["c1_k16_idx"])
    ok = golden["c1_k16_pca_planarity"] < 0.2          # uniform volume data: most neighbourhoods are NOT planar; compare the well-conditioned ones
    assert ok.any() and np.abs(got[ok] - golden["c1_k16_pca_nrm"][ok]).max() <= 1e-4


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_pca_normals(pt, oracle, dtype):
    """fp16 clouds stay fp16 in the resident input; the PCA table is packed from their fp32 image (made on demand)."""
    rng = np.random.default_rng(5)
    n = 50000
    src = rng.random((3, n)).astype(np.float32)
    src[2] = (0.3 + 0.1 * src[0] + 0.05 * np.sin(6 * src[1]) + 1e-3 * rng.standard_normal(n)).astype(np.float32)   # a noisy surface
    src = src.astype(dtype)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
    tgt = src[:, :3000].copy()
    pt.build(src, None, nrm)
    idx, d2 = pt.query(tgt, 16)
    _check_exact((idx, d2), oracle.knn_bruteforce(src.astype(np.float64), tgt.astype(np.float64), 16), "surface cloud")
    got = pt.pca_normals(idx)
    src = src.astype(np.float32)
    want, plan = oracle.pca_normals(idx, src, nrm)
    ok = plan < 0.1                                  # well-conditioned neighbourhoods
    assert ok.mean() > 0.9
    assert np.abs(got[ok] - want[ok]).max() <= TOL
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-5 and (got[ok][:, 2] > 0).all()


# ---- multi-GPU pieces on one device: G logical slabs through the same kernels ---------------------------------
@pytest.mark.parametrize("g", [2, 4])
def test_logical_slabs_merge_equals_global(pkg, oracle, g):
    import torch
    n, m, k, seed = 120000, 6000, 8, 0xC4
    src, tgt = oracle.synth_xyz(seed, 0, n), oracle.synth_xyz(seed, 1, m)
    want = oracle.KdTree(src).query(tgt, k)
    bounds = np.concatenate([[-math.inf], np.quantile(src[0], np.arange(1, g) / g), [math.inf]])
    li = torch.empty((g, m, k), dtype=torch.int32, device="cuda"); ld = torch.empty((g, m, k), dtype=torch.float64, device="cuda")
    tx = torch.from_numpy(tgt).cuda()
    ctxs = []
    for s in range(g):
        p = pkg.PointsTransfer(device=0)
        p.build_synth(n, seed, slab_axis=0, slab_lo=bounds[s], slab_hi=bounds[s + 1])
        ctxs.append(p)
    assert sum(p.num_source for p in ctxs) == n
    for s, p in enumerate(ctxs):            # v0: every slab answers every target, then one G-way merge
        p.query_dev(tx, pkg.F32, m, k, li[s], ld[s])
    oi = torch.empty((m, k), dtype=torch.int32, device="cuda"); od = torch.empty((m, k), dtype=torch.float64, device="cuda")
    ctxs[0].merge_candidates_dev(li, ld, g, m, k, oi, od)
    torch.cuda.synchronize()
    _check_exact((oi.cpu().numpy().view(np.uint32), od.cpu().numpy()), want, "v0 merge g=%d" % g)
    # v2: home slab first, then only the slabs the need-mask names, bounded by the current k-th distance
    home = np.clip(np.searchsorted(bounds, tgt[0], side="right") - 1, 0, g - 1)
    fi = np.full((m, k), 0xFFFFFFFF, np.uint32); fd = np.full((m, k), np.inf)
    total_foreign = 0
    for s, p in enumerate(ctxs):
        mine = np.nonzero(home == s)[0]
        ms = len(mine)
        txs = torch.from_numpy(np.ascontiguousarray(tgt[:, mine])).cuda()
        hi_ = torch.empty((ms, k), dtype=torch.int32, device="cuda"); hd_ = torch.empty((ms, k), dtype=torch.float64, device="cuda")
        p.query_dev(txs, pkg.F32, ms, k, hi_, hd_)
        need = torch.empty((g, ms), dtype=torch.uint8, device="cuda")
        p.slab_need_dev(txs, pkg.F32, hd_, ms, k, 0, bounds, s, need)
        torch.cuda.synchronize()
        need_h = need.cpu().numpy()
        assert not need_h[s].any()
        lists_i = [hi_]; lists_d = [hd_]
        for s2, p2 in enumerate(ctxs):
            sel = np.nonzero(need_h[s2])[0]
            ci = torch.full((ms, k), -1, dtype=torch.int32, device="cuda"); cd = torch.full((ms, k), math.inf, dtype=torch.float64, device="cuda")
            if s2 != s and len(sel):
                total_foreign += len(sel)
                sx = torch.from_numpy(np.ascontiguousarray(tgt[:, mine[sel]])).cuda()
                bnd = hd_[torch.from_numpy(sel).cuda(), k - 1].contiguous()
                ri = torch.empty((len(sel), k), dtype=torch.int32, device="cuda"); rd = torch.empty((len(sel), k), dtype=torch.float64, device="cuda")
                p2.query_bounded_dev(sx, pkg.F32, bnd, len(sel), k, ri, rd)
                ci[torch.from_numpy(sel).cuda()] = ri; cd[torch.from_numpy(sel).cuda()] = rd
            if s2 != s:
                lists_i.append(ci); lists_d.append(cd)
        Li = torch.stack(lists_i).contiguous(); Ld = torch.stack(lists_d).contiguous()
        mi = torch.empty((ms, k), dtype=torch.int32, device="cuda"); md = torch.empty((ms, k), dtype=torch.float64, device="cuda")
        p.merge_candidates_dev(Li, Ld, g, ms, k, mi, md)
        torch.cuda.synchronize()
        fi[mine] = mi.cpu().numpy().view(np.uint32); fd[mine] = md.cpu().numpy()
    _check_exact((fi, fd), want, "v2 pruned g=%d" % g)
    assert total_foreign < 0.25 * m * (g - 1)           # the pruning really prunes
    for p in ctxs:
        p.close()


def test_fp16_slabs_of_the_clustered_generator(pkg, oracle):
    """Config 5 as its multi-GPU run builds it: every slab generates its own part of the clustered fp16 cloud (fp16-resident input
    with global indices), answers every target, and the G-way merge equals the oracle over the whole cloud."""
    import torch
    g, n, m, k, seed = 3, 150000, 4000, 32, 0xC5
    src = oracle.synth_xyz(seed, 0, n, dist=1, n_total=n, m_total=m).astype(np.float16).astype(np.float32)
    tgt = oracle.synth_xyz(seed, 1, m, dist=1, n_total=n, m_total=m).astype(np.float16).astype(np.float32)
    want = oracle.KdTree(src.astype(np.float64)).query(tgt.astype(np.float64), k)
    bounds = np.concatenate([[-math.inf], np.quantile(src[0], np.arange(1, g) / g), [math.inf]])
    li = torch.empty((g, m, k), dtype=torch.int32, device="cuda"); ld = torch.empty((g, m, k), dtype=torch.float64, device="cuda")
    tx = torch.from_numpy(tgt).cuda()
    ctxs = []
    for s in range(g):
        p = pkg.PointsTransfer(device=0, k_hint=k)
        p.build_synth(n, seed, dist=pkg.capi.DIST_CLUSTERED, xyz_type=pkg.F16, slab_axis=0, slab_lo=bounds[s], slab_hi=bounds[s + 1])
        ctxs.append(p)
    assert sum(p.num_source for p in ctxs) == n
    for s, p in enumerate(ctxs):
        p.query_dev(tx, pkg.F32, m, k, li[s], ld[s])
    oi = torch.empty((m, k), dtype=torch.int32, device="cuda"); od = torch.empty((m, k), dtype=torch.float64, device="cuda")
    ctxs[0].merge_candidates_dev(li, ld, g, m, k, oi, od)
    torch.cuda.synchronize()
    _check_exact((oi.cpu().numpy().view(np.uint32), od.cpu().numpy()), want, "fp16 slabs")
    for p in ctxs:
        p.close()


# ---- BASELINE config 2 at full size: size-independent properties + sampled exactness ----------------------------
def test_full_size_c2_properties(pkg, oracle):
    import torch
    n, m, k, seed = 10_000_000, 1_000_000, 8, 0xC2
    with pkg.PointsTransfer(device=0) as p:
        p.build_synth(n, seed)
        p.targets_synth(m, seed)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        st = p.stats()
        idx2 = torch.empty_like(idx); d22 = torch.empty_like(d2)
        p.rebuild(); p.query_resident_dev(k, idx2, d22)                    # idempotence: rebuild + re-query is bit-identical
        torch.cuda.synchronize()
        assert torch.equal(idx, idx2) and torch.equal(d2, d22)
    I = idx.cpu().numpy().view(np.uint32); D = d2.cpu().numpy()
    assert I.max() < n and (np.diff(D, axis=1) >= 0).all()                  # valid, ascending
    tie = np.diff(D, axis=1) == 0
    assert (np.diff(I.astype(np.int64), axis=1)[tie] > 0).all()             # ties broken by index
    assert (np.sort(I, axis=1)[:, 1:] != np.sort(I, axis=1)[:, :-1]).all()  # no neighbour twice
    src, tgt = oracle.synth_xyz(seed, 0, n), oracle.synth_xyz(seed, 1, m)
    # returned distances are the metric of the returned indices (checksum over all rows)
    dx = tgt[0].astype(np.float64)[:, None] - src[0][I]; dy = tgt[1].astype(np.float64)[:, None] - src[1][I]; dz = tgt[2].astype(np.float64)[:, None] - src[2][I]
    assert np.array_equal((dx * dx + dy * dy) + dz * dz, D)
    # exactness on a sample, against the CPU kd-tree restatement over the full cloud
    sel = np.random.default_rng(0).choice(m, 20000, replace=False)
    wi, wd = oracle.KdTree(src).query(tgt[:, sel], k)
    assert np.array_equal(I[sel], wi) and np.array_equal(D[sel], wd)
    assert st["n_source"] == n and st["n_target"] == m and st["ms_query"] > 0


def test_full_size_c3_sampled_exactness(pkg, oracle):
    """BASELINE config 3 at full size (100M / 10M / k=16): valid, ascending, tie-broken, and exact on a sample against the
    CPU kd-tree restatement built over the whole cloud."""
    import torch
    n, m, k, seed = 100_000_000, 10_000_000, 16, 0xC3
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build_synth(n, seed)
        p.targets_synth(m, seed)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        assert bool((d2[:, 1:] >= d2[:, :-1]).all())
        sel = torch.from_numpy(np.random.default_rng(1).choice(m, 5000, replace=False)).cuda()
        I = idx[sel].cpu().numpy().view(np.uint32); D = d2[sel].cpu().numpy()
        pn = torch.empty((m, 3), dtype=torch.float32, device="cuda")
        p.pca_normals_dev(idx, m, k, pn)                                     # config 3's PCA normals over ALL 10M targets
        torch.cuda.synchronize()
        PN = pn[sel].cpu().numpy()
        assert bool(((pn.norm(dim=1) - 1).abs() < 1e-4).all())
    src = oracle.synth_xyz(seed, 0, n)
    tgt = oracle.synth_xyz(seed, 1, m)[:, sel.cpu().numpy()]
    wi, wd = oracle.KdTree(src).query(tgt, k)
    assert np.array_equal(I, wi) and np.array_equal(D, wd)
    # the PCA of the sampled rows against the oracle (full-size cloud, the generator's stored normals orient the sign);
    # uniform volume data is mostly NOT planar, so -- as in test_pca_matches_golden -- the well-conditioned rows are compared
    want, plan = oracle.pca_normals(wi, src, oracle.synth_nrm(seed, n))
    ok = plan < 0.2
    assert ok.sum() > 100 and np.abs(PN[ok] - want[ok]).max() <= 1e-4


def test_full_size_c4_two_kernels_agree(pkg):
    """BASELINE config 4 at full size (1B / 50M / k=8) is beyond what the CPU oracle can hold, so the check is structural:
    results are valid / ascending / duplicate-free, a rebuild reproduces them bit for bit, and on a 200k-target sample the
    tile kernel and the independent group kernel return identical neighbours and distances."""
    import torch
    n, m, k, seed = 1_000_000_000, 50_000_000, 8, 0xC4
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build_synth(n, seed)
        p.targets_synth(m, seed)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        assert p.stats()["n_leftover"] < 0.01 * m
        assert bool((d2[:, 1:] >= d2[:, :-1]).all()) and bool((idx.view(torch.int32) != -1).all())
        srt = idx.to(torch.int64).sort(dim=1).values
        assert bool((srt[:, 1:] != srt[:, :-1]).all())                       # no neighbour twice
        chk = (idx.to(torch.int64) & 0xFFFFFFFF).sum().item(), d2.sum().item()
        p.rebuild()
        i2 = torch.empty_like(idx); e2 = torch.empty_like(d2)
        p.query_resident_dev(k, i2, e2)
        torch.cuda.synchronize()
        assert ((i2.to(torch.int64) & 0xFFFFFFFF).sum().item(), e2.sum().item()) == chk and torch.equal(i2, idx)
        del i2, e2, srt
        xyz = torch.empty((3, m), dtype=torch.float32, device="cuda")
        p.resident_target_xyz_dev(xyz)
        sel = torch.randperm(m, device="cuda")[:200_000]
        sx = xyz[:, sel].contiguous()
        p.set_param("tile", 0)                                               # group kernel only
        gi = torch.empty((200_000, k), dtype=torch.int32, device="cuda"); gd = torch.empty((200_000, k), dtype=torch.float64, device="cuda")
        p.query_dev(sx, pkg.F32, 200_000, k, gi, gd)
        torch.cuda.synchronize()
        assert p.stats()["n_leftover"] == 0
        assert torch.equal(gi, idx[sel]) and torch.equal(gd, d2[sel])


def test_full_size_c4_against_the_oracle_in_sub_boxes(pkg, oracle):
    """BASELINE config 4 at full size against an answer that does not depend on the grid: the oracle streams the generator
    over all 1e9 source indices on the host (never holding the cloud) and keeps the points inside 64 small boxes; the targets
    inside the INNER part of each box are brute-forced against them.  Every such k-th distance is smaller than the margin
    between inner and outer box, so no point outside the outer box can be (or tie with) a neighbour: the brute-force answer
    IS the answer over the whole cloud, and the GPU rows must equal it bit for bit (indices, distances) -- a build that lost,
    duplicated or misplaced records in these regions cannot pass.  Also: the fused blend of those rows within 1e-5."""
    import torch
    n, m, k, seed = 1_000_000_000, 50_000_000, 8, 0xC4
    a, g = np.float32(0.008), np.float32(0.004)              # inner side, margin (k-th distance at this density: ~1.2e-3)
    xs = np.array([0.0, 0.3137, 0.62, 1.0 - 0.008], np.float32)              # four x-slabs (both faces of the cube among them)
    yz = np.array([0.0, 0.2501, 0.5003, 1.0 - 0.008], np.float32)
    inner_lo = np.array([[x, y, z] for x in xs for y in yz for z in yz], np.float32)
    inner_hi = inner_lo + a
    outer_lo, outer_hi = inner_lo - g, inner_hi + g
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build_synth(n, seed)
        p.targets_synth(m, seed)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        rgb = torch.empty((m, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((m, 3), dtype=torch.float32, device="cuda")
        p.query_blend_resident_dev(k, pkg.BLEND_MEAN, idx, d2, rgb, nrm)
        torch.cuda.synchronize()
        sxyz, sidx, sbox = oracle.synth_filter_boxes(seed, 0, n, outer_lo, outer_hi)
        txyz, tidx, tbox = oracle.synth_filter_boxes(seed, 1, m, inner_lo, inner_hi)
        assert len(tidx) > 500 and len(sidx) > 64 * 1000
        rows = torch.from_numpy(tidx.astype(np.int64)).cuda()
        I = idx[rows].cpu().numpy().view(np.uint32); D = d2[rows].cpu().numpy()
        C = rgb[rows].cpu().numpy(); N = nrm[rows].cpu().numpy()
    checked = 0
    for b in range(len(inner_lo)):
        ss, tt = np.nonzero(sbox == b)[0], np.nonzero(tbox == b)[0]
        if not len(tt):
            continue
        wi, wd = oracle.knn_bruteforce(sxyz[:, ss], txyz[:, tt], k, gidx=sidx[ss])
        assert (wd[:, k - 1] < float(g) * float(g)).all()                   # the margin argument holds for every row
        assert np.array_equal(I[tt], wi) and np.array_equal(D[tt], wd), "box %d" % b
        checked += len(tt)
    assert checked == len(tidx)
    # fused blend of the checked rows: attributes are index-addressable too
    flat = I.reshape(-1).astype(np.int64)
    uniq, inv = np.unique(flat, return_inverse=True)
    arg = np.empty((len(uniq), 3), np.uint8); anr = np.empty((len(uniq), 3), np.float32)
    for j, u in enumerate(uniq):
        arg[j] = oracle.synth_rgb(seed, 1, i0=int(u))[0]; anr[j] = oracle.synth_nrm(seed, 1, i0=int(u))[0]
    rc, rn = oracle.blend(inv.reshape(I.shape).astype(np.uint32), D, arg, anr, mode=0)
    assert np.abs(C - rc).max() / 255.0 <= TOL and np.abs(N - rn).max() <= TOL


def test_slab_engine_with_no_local_targets(pkg, oracle):
    """A rank whose slab holds no targets still takes part in the exchange: its request packing returns empty tensors
    (it used to raise before reaching the collective, leaving the other ranks blocked)."""
    import torch
    from pt_amd import sharding
    src = oracle.synth_xyz(3, 0, 4000)
    with pkg.PointsTransfer(device=0) as p:
        p.build(src)
        eng = sharding.GpuSlabEngine(p, pkg.F32, torch.device("cuda", 0))
        xyz = torch.empty((3, 0), dtype=torch.float32, device="cuda"); d2 = torch.empty((0, 8), dtype=torch.float64, device="cuda")
        sel, pkt = eng.pack_requests(xyz, d2, 8, 0, sharding.uniform_slab_bounds(2), 0)
        assert sel.numel() == 0 and tuple(pkt.shape) == (0, 5)
        idx = torch.empty((0, 8), dtype=torch.int32, device="cuda")
        st = sharding.exchange_and_merge(sharding.SingleComm(), eng, xyz, idx, d2, 8, 0, sharding.uniform_slab_bounds(1))
        assert st["crossing"] == 0


def test_even_chunk_scatter_variant_against_oracle(pkg, oracle):
    """Pass 1 of the build switches to 1024-thread workgroups (8192-record tiles) once a cloud has an even number of tiles
    per chunk -- from 33.5 M points up.  40 M points: the smallest oracle-checked cloud that takes that variant."""
    n, m, k, seed = 40_000_000, 20_000, 8, 0x40
    src = oracle.synth_xyz(seed, 0, n)
    tgt = oracle.synth_xyz(seed, 1, m)
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build(src)
        assert p.stats()["pass1_pooled"] == 1          # (from 32 Mi points up pass 1 sizes its bins from a sample: no histogram pass)
        idx, d2 = p.query(tgt, k)
        p.set_param("pool_min_points", 0)              # the exact, histogram-first pass 1 on the same cloud
        p.rebuild()
        assert p.stats()["pass1_pooled"] == 0
        idx0, d20 = p.query(tgt, k)
    wi, wd = oracle.KdTree(src).query(tgt, k)
    assert np.array_equal(idx, wi) and np.array_equal(d2, wd)
    assert np.array_equal(idx0, wi) and np.array_equal(d20, wd)


@pytest.mark.parametrize("f64,slab", [(False, False), (True, False), (False, True)])
def test_pooled_pass1_on_small_clouds(pkg, oracle, f64, slab):
    """The pooled pass 1 (bin regions from a sample, space taken in blocks, sentinel padding dropped by pass 2) forced onto clouds the
    oracle answers in seconds: fp32, fp64 (32-byte records) and a slab with caller-given global indices; blobs make the bins uneven."""
    rng = np.random.default_rng(77)
    n, m, k = 3_000_000, 4000, 8
    src = rng.random((3, n))
    src[:, : n // 3] = 0.5 + 0.02 * rng.standard_normal((3, n // 3))            # a third of the cloud in one blob
    src = np.clip(src, 0.0, 1.0)
    src = src if f64 else src.astype(np.float32)
    tgt = src[:, rng.integers(0, n, m)] + (1e-4 * rng.standard_normal((3, m))).astype(src.dtype)
    gidx = rng.permutation(n).astype(np.uint32) if slab else None
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.set_param("pool_min_points", 1)
        p.build(src, gidx=gidx)
        st = p.stats()
        assert st["n_levels"] == 2 and st["pass1_pooled"] == 1, st
        idx, d2 = p.query(tgt, k)
        p.rebuild()                                                             # and again from the remembered grid
        assert p.stats()["pass1_pooled"] == 1
        idx2, d22 = p.query(tgt, k)
    wi, wd = oracle.KdTree(src.astype(np.float64)).query(tgt.astype(np.float64), k)
    if slab:
        wi = np.where(wi == pkg.NOIDX, wi, gidx[np.minimum(wi, n - 1)])
        order = np.lexsort((wi, wd), axis=1)                                    # ties are broken by the GLOBAL index
        wi = np.take_along_axis(wi, order, 1); wd = np.take_along_axis(wd, order, 1)
    assert np.array_equal(idx, wi) and np.array_equal(d2, wd)
    assert np.array_equal(idx2, wi) and np.array_equal(d22, wd)


@pytest.mark.parametrize("f64,pool1,slab", [(False, False, False), (True, False, False), (False, True, False), (False, True, True)])
def test_pooled_pass2_on_a_uniform_cloud(pkg, oracle, f64, pool1, slab):
    """A rebuild of a resident cloud that the previous build found uniform sizes the blocks of pass 2 from the macro counts (no pass-2
    histogram, no block ids beside the records): exact like the first build, with the exact and with the pooled pass 1 before it."""
    rng = np.random.default_rng(91)
    n, m, k = 3_000_000, 4000, 8
    src = rng.random((3, n)) if f64 else rng.random((3, n), dtype=np.float32)
    tgt = rng.random((3, m)).astype(src.dtype)
    gidx = rng.permutation(n).astype(np.uint32) if slab else None            # a slab of a larger cloud: caller-given global indices
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        if pool1:
            p.set_param("pool_min_points", 1)
        p.build(src, gidx=gidx)
        st = p.stats()
        # (round 4: a cloud big enough for the pooled pass 1 -- "pool_min_points", lowered here -- is asked through a 1/64 sample BEFORE its
        #  first sort and pools pass 2 on that FIRST build; smaller clouds learn it from their first build's occupancy, as before)
        assert st["n_levels"] == 2 and st["pass2_pooled"] == (1 if pool1 else 0) and st["uniform_probe"] == (1 if pool1 else 0) and st["pass1_pooled"] == (1 if pool1 else 0), st
        i0, d0 = p.query(tgt, k)
        p.rebuild()
        st = p.stats()
        assert st["pass2_pooled"] == 1 and st["pass1_pooled"] == (1 if pool1 else 0), st
        i1, d1 = p.query(tgt, k)
        p.set_param("pool2", 0)
        p.rebuild()
        assert p.stats()["pass2_pooled"] == 0
        i2, d2 = p.query(tgt, k)
    wi, wd = oracle.KdTree(src.astype(np.float64)).query(tgt.astype(np.float64), k)
    if slab:
        wi = gidx[wi]
        order = np.lexsort((wi, wd), axis=1)
        wi = np.take_along_axis(wi, order, 1); wd = np.take_along_axis(wd, order, 1)
    for gi, gd in ((i0, d0), (i1, d1), (i2, d2)):
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd)


def test_pooled_pass2_overflow_falls_back(pkg, oracle):
    """A cloud whose cells are all occupied at about rho points -- so the build calls it uniform -- but whose density changes several-fold
    across every macro block: the blocks' regions, sized as if the macro were uniform inside, overflow; the flag sends the rebuild back to
    the exact pass 2 and the context stops pooling pass 2 for this cloud."""
    rng = np.random.default_rng(92)
    n, m, k = 3_000_000, 3000, 8
    src = rng.random((3, n), dtype=np.float32)
    src[0] = (0.15 + 0.85 * src[0]) ** 2.0                                   # density ~ 1 / sqrt(x): 2.6 x from one end to the other, no empty region
    tgt = rng.random((3, m), dtype=np.float32); tgt[0] = (0.15 + 0.85 * tgt[0]) ** 2.0
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.set_param("adaptive", 1)
        p.build(src)
        st = p.stats()
        if not (st["n_refine"] == 0 and st["rho_occupied"] <= 1.25 * 4.0):
            pytest.skip("the build refined this cloud (rho_occupied %.2f): it never pools pass 2" % st["rho_occupied"])
        p.rebuild()
        assert p.stats()["pass2_pooled"] == -1, p.stats()
        idx, d2 = p.query(tgt, k)
        p.rebuild()
        assert p.stats()["pass2_pooled"] == 0
    wi, wd = oracle.KdTree(src).query(tgt, k)
    assert np.array_equal(idx, wi) and np.array_equal(d2, wd)


def test_uniformity_sample_decides_the_first_build(pkg, oracle):
    """What the 1/64 sample taken before a cloud's FIRST sort says (pt_stats.uniform_probe) and what the build does with it: a uniform cloud
    pools pass 2 at once; a smooth density gradient -- no block far off, the chi-square sum is -- and a clump are refused (exact pass 2, no
    build wasted on a region that overflows); "forget" makes a rebuild ask again.  Same neighbours as the oracle every time."""
    rng = np.random.default_rng(93)
    n, m, k = 3_000_000, 3000, 8
    uni = rng.random((3, n), dtype=np.float32)
    grad = uni.copy(); grad[0] = (0.15 + 0.85 * grad[0]) ** 2.0                 # density ~ 1 / sqrt(x): 2.6 x across the cloud, every cell occupied
    clump = uni.copy(); clump[:, : n // 50] = (0.4 + 0.01 * rng.random((3, n // 50))).astype(np.float32)
    tgt = rng.random((3, m), dtype=np.float32)
    for name, src, want_probe in (("uniform", uni, 1), ("gradient", grad, -1), ("clump", clump, -1)):
        with pkg.PointsTransfer(device=0, k_hint=k) as p:
            p.set_param("pool_min_points", 1)
            p.build(src)
            st = p.stats()
            assert st["n_levels"] == 2 and st["uniform_probe"] == want_probe, (name, st)
            assert st["pass2_pooled"] == (1 if want_probe == 1 else 0), (name, st)          # never -1: the sample kept the overflow from happening
            got = p.query(tgt, k)
            p.rebuild()
            assert p.stats()["uniform_probe"] == 0, name                                    # a finished build knows: no second sample
            p.set_param("forget", 1)
            p.rebuild()
            assert p.stats()["uniform_probe"] == want_probe, name
            got2 = p.query(tgt, k)
        want = oracle.KdTree(src).query(tgt, k)
        _check_exact(got, want, name)
        _check_exact(got2, want, name + " after forget")


def test_first_build_takes_its_cell_size_from_the_sample(pkg, oracle):
    """Round 4: a detail-transfer run builds its index ONCE, so the first build is the one that counts.  The sample taken before a cloud's first
    sort also estimates the points per occupied cell (distinct cells seen per block -> cells occupied), and a cloud it finds far from uniform --
    a surface -- gets the finer cell size before that sort instead of after it: one sort, not two ("presort_refine" = 0: round 3's behaviour).
    A cloud stored in SPATIAL order gives itself away in the same sample (consecutive points share a block): nothing is taken from the sample
    then -- exact passes, the sort's own count -- where the pooled pass 1 used to overflow and be redone.  Same neighbours as the oracle."""
    rng = np.random.default_rng(97)
    n, m, k = 3_000_000, 3000, 8
    v = rng.standard_normal((3, n)).astype(np.float32); v /= np.linalg.norm(v, axis=0, keepdims=True)
    shell = (0.5 + 0.45 * v + 1e-4 * rng.standard_normal((3, n)).astype(np.float32)).astype(np.float32)
    uni = rng.random((3, n), dtype=np.float32)
    key = (uni[0] * 32).astype(np.int32) * 1024 + (uni[1] * 32).astype(np.int32) * 32 + (uni[2] * 32).astype(np.int32)
    ordered = np.ascontiguousarray(uni[:, np.argsort(key, kind="stable")])
    t_shell = (shell[:, :m] + np.float32(1e-3)).astype(np.float32)
    t_uni = rng.random((3, m), dtype=np.float32)
    seen = {}
    for name, src, tgt in (("shell", shell, t_shell), ("ordered", ordered, t_uni), ("uniform", uni, t_uni)):
        want = oracle.KdTree(src).query(tgt, k)
        for pre in (1, 0):
            with pkg.PointsTransfer(device=0, k_hint=k) as p:
                p.set_param("pool_min_points", 1)
                p.set_param("presort_refine", pre)
                p.build(src)
                st = p.stats()
                seen[(name, pre)] = st
                _check_exact(p.query(tgt, k), want, "%s presort=%d" % (name, pre))
                p.rebuild()                                                     # the finished build left its cell size behind: one sort, nothing asked
                st2 = p.stats()
                assert st2["n_sorts"] == 1 and st2["presort_refine"] == 0 and st2["uniform_probe"] == 0, (name, pre, st2)
                assert st2["grid_dim"][:] == st["grid_dim"][:], (name, st["grid_dim"][:], st2["grid_dim"][:])
                p.set_param("forget", 1); p.rebuild()
                st3 = p.stats()
                assert (st3["n_sorts"], st3["presort_refine"], st3["ordered_input"]) == (st["n_sorts"], st["presort_refine"], st["ordered_input"]), (name, pre, st3)
                _check_exact(p.query(tgt, k), want, "%s presort=%d after forget" % (name, pre))
    a, b = seen[("shell", 1)], seen[("shell", 0)]
    assert a["uniform_probe"] == -1 and a["presort_refine"] >= 1 and a["n_sorts"] == 1 and a["ordered_input"] == 0, a
    assert b["presort_refine"] == 0 and b["n_sorts"] >= 2 and b["n_refine"] >= 1, b
    assert a["rho_occupied"] <= 1.3 * b["rho_occupied"], (a["rho_occupied"], b["rho_occupied"])      # as fine a grid as the counted refinement reaches
    for pre in (1, 0):
        o = seen[("ordered", pre)]
        assert o["ordered_input"] == 1 and o["n_sorts"] == 1 and o["pass1_pooled"] == 0 and o["pass2_pooled"] == 0 and o["presort_refine"] == 0, o
        u = seen[("uniform", pre)]
        assert u["ordered_input"] == 0 and u["uniform_probe"] == 1 and u["n_sorts"] == 1 and u["presort_refine"] == 0 and u["pass2_pooled"] == 1, u


@pytest.mark.parametrize("cpp", [2, 8])
def test_a_sample_that_lies_about_the_occupancy_is_overruled_by_the_count(pkg, oracle, cpp):
    """The first build's sample is every 16th run of 256 consecutive points.  Here exactly those runs sit in a small cube and everything else
    is uniform: the sample says "a clump, few cells occupied -- refine", the sampled bounding box is wrong as well, and the pooled regions are
    sized for a clump.  Every guess is verified by the build that uses it: the box is found wrong and the build starts over on the exact one,
    the sort's own count of occupied cells bounds what the sample's refinement may leave behind -- with the usual limit of two cells per point
    the refined grid is fine but legal, with eight (cpp = 8) its cells hold a fraction of rho and the sample's answer is thrown away
    (presort_refine < 0: grid from the box, exact passes) -- and the search is the oracle's."""
    rng = np.random.default_rng(98)
    n, m, k = 3_000_000, 3000, 8
    src = rng.random((3, n), dtype=np.float32)
    run = np.arange(n) // 256
    lied = (run % 16) == 0
    src[:, lied] = (0.4 + 0.2 * rng.random((3, int(lied.sum())))).astype(np.float32)
    tgt = rng.random((3, m), dtype=np.float32)
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.set_param("pool_min_points", 1)
        p.set_param("guess_min_points", 100000)
        p.set_param("refine_cells_per_point", cpp)
        p.build(src)
        st = p.stats()
        got = p.query(tgt, k)
        keep = {f: st[f] for f in ("uniform_probe", "ordered_input", "presort_refine", "n_sorts", "bbox_guess", "rho_occupied", "n_refine", "pass1_pooled")}
        assert st["bbox_guess"] == -1 and st["n_sorts"] >= 2 and st["ordered_input"] == 0, keep      # the sampled box was wrong: built again on the exact one
        assert st["uniform_probe"] == -1, keep                                      # ... where the sample no longer looks uniform
        assert 0.4 * 4.0 <= st["rho_occupied"] <= 8.0, keep                         # the count had the last word
        if cpp == 8:
            assert st["presort_refine"] < 0 and st["n_sorts"] >= 3 and st["pass1_pooled"] == 0, keep
            assert st["rho_occupied"] >= 3.0, keep                                  # the grid the cloud wants, from the box
        p.rebuild()
        st2 = p.stats()
        assert st2["n_sorts"] == 1 and st2["presort_refine"] == 0, st2
        got2 = p.query(tgt, k)
    want = oracle.KdTree(src).query(tgt, k)
    _check_exact(got, want, "sample lied")
    _check_exact(got2, want, "sample lied, rebuild")


def test_wrong_sampled_box_and_pool_overflow_in_one_build(pkg, oracle):
    """Two guesses of one build fail together: the sampled bounding box misses an outlier AND the pooled pass 1's sampled bin regions
    overflow (every point outside the sampled runs sits in one clump).  rebuild() restarts itself once per failed guess; the result is
    the exact build's, and the context remembers both failures."""
    rng = np.random.default_rng(94)
    n, m, k = 3_000_000, 3000, 8
    src = rng.random((3, n), dtype=np.float32)
    run = np.arange(n) // 256
    hidden = (run % 64) != 0                                  # the pooled pass 1 samples one run of 256 in 64 ...
    src[:, hidden] = (0.25 + 0.01 * rng.random((3, int(hidden.sum())))).astype(np.float32)
    src[:, 300] = (3.0, -2.0, 7.5)                           # ... and the bounding box one run in 1024: this point is in neither sample
    tgt = np.concatenate([rng.random((3, m // 2), dtype=np.float32), (0.25 + 0.01 * rng.random((3, m - m // 2))).astype(np.float32),
                          np.array([[2.9], [-1.9], [7.4]], np.float32)], axis=1)
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.set_param("pool_min_points", 1)
        p.set_param("guess_min_points", 100000)
        p.set_param("adaptive", 0)
        p.build(src)
        st = p.stats()
        assert st["bbox_guess"] == -1, st
        got = p.query(tgt, k)
        p.rebuild()
        st2 = p.stats()
        assert st2["bbox_guess"] == 0 and st2["pass1_pooled"] in (0, 1), st2
        got2 = p.query(tgt, k)
    want = oracle.KdTree(src).query(tgt, k)
    _check_exact(got, want, "both guesses failed")
    _check_exact(got2, want, "rebuild after both failures")


@pytest.mark.parametrize("at", [0.25, 0.985])
def test_pooled_pass1_overflow_falls_back_to_the_exact_histogram(pkg, oracle, at):
    """A cloud built to fool the sample: the runs the sample reads (one run of 256 points in 64) are uniform, every other point sits in
    one small clump.  The clump's bin outgrows the region its sample gave it, the flag is raised, and the build is redone with the
    exact pass 1 -- same answer as the oracle, and the context stops pooling for this cloud.  at = 0.985: the clump sits in the LAST
    macro block, whose region ends where the records array does (the passes downstream of a failed guess must not touch the
    reserved-but-unwritten stretch, let alone write past the allocation: they return at once, pt_grid.hip scatter_kernel)."""
    rng = np.random.default_rng(5)
    n, m, k = 3_000_000, 3000, 8
    src = rng.random((3, n), dtype=np.float32)
    run = np.arange(n) // 256
    hidden = (run % 64) != 0
    src[:, hidden] = (at + 0.01 * rng.random((3, int(hidden.sum())))).astype(np.float32)
    tgt = np.concatenate([rng.random((3, m // 2), dtype=np.float32), (at + 0.01 * rng.random((3, m - m // 2))).astype(np.float32)], axis=1)
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.set_param("pool_min_points", 1)
        p.set_param("adaptive", 0)                         # (keep the coarse grid of the bounding box: the clump stays in one macro bin)
        p.build(src)
        st = p.stats()
        assert st["pass1_pooled"] == -1, st
        idx, d2 = p.query(tgt, k)
        p.rebuild()
        assert p.stats()["pass1_pooled"] == 0              # no more pooling for this cloud
    wi, wd = oracle.KdTree(src).query(tgt, k)
    assert np.array_equal(idx, wi) and np.array_equal(d2, wd)


def test_nan_coordinates(pkg, pt, oracle):
    """A NaN source coordinate is an argument error (the bounding-box reductions must not swallow it); a NaN TARGET coordinate
    gives that row PT_NOIDX / +inf in every slot and leaves the other rows alone."""
    rng = np.random.default_rng(11)
    src = rng.random((3, 5000), dtype=np.float32)
    bad = src.copy(); bad[1, 1234] = np.nan
    with pytest.raises(pkg.PtError) as e:
        pt.build(bad)
    assert e.value.code == pkg.capi.ERR_ARG
    big = rng.random((3, 9_000_000), dtype=np.float32)                      # sampled-bounding-box path: pass 1 verifies the box
    big[2, 8_765_432] = np.nan
    with pytest.raises(pkg.PtError) as e:
        pt.build(big)
    assert e.value.code == pkg.capi.ERR_ARG
    pt.build(src)
    tgt = rng.random((3, 300), dtype=np.float32)
    tgt[0, 7] = np.nan; tgt[2, 200] = np.nan
    idx, d2 = pt.query(tgt, 8)
    wi, wd = oracle.knn_bruteforce(src, tgt, 8)
    good = np.ones(300, bool); good[[7, 200]] = False
    assert np.array_equal(idx[good], wi[good]) and np.array_equal(d2[good], wd[good])
    assert (idx[~good] == pkg.NOIDX).all() and np.isinf(d2[~good]).all()


def test_cell_side_beyond_fp32_range(pkg, oracle):
    """Coordinates scaled by 1e20: the squared cell side leaves fp32's range, where the tile kernel's fp32 pruning would
    skip neighbouring cells -- such clouds must be answered by the fp64-pruning group kernel, bit-exact as ever."""
    rng = np.random.default_rng(12)
    src = (rng.random((3, 20000)) * 1e20).astype(np.float32)
    tgt = (rng.random((3, 500)) * 1e20).astype(np.float32)
    with pkg.PointsTransfer(device=0) as p:
        p.build(src)
        idx, d2 = p.query(tgt, 8)
    wi, wd = oracle.knn_bruteforce(src, tgt, 8)
    assert np.array_equal(idx, wi) and np.array_equal(d2, wd)


@pytest.mark.parametrize("scale", [1e-19, 1e-12, 1e18])
def test_wave_kernel_fp32_prefilter_at_extreme_scales(pkg, oracle, scale):
    """The wave kernel's fp32 pre-filter where fp32 squares underflow (1e-19: d2 ~ 1e-40, denormal or zero in fp32) or overflow
    (1e18: d2 ~ 1e36 - 1e38, next to FLT_MAX): it may only ever let MORE through than the exact test, never less."""
    rng = np.random.default_rng(int(-np.log10(scale)) + 40)
    base = rng.random((3, 30000))
    base[:, 5000:9000] = base[:, [77]] + 1e-3 * rng.standard_normal((3, 4000))      # a clump (dense cell: long scans, many candidates)
    src = (base * scale).astype(np.float32)
    tgt = np.concatenate([src[:, 5000:5300], (rng.random((3, 300)) * scale).astype(np.float32)], axis=1)
    with pkg.PointsTransfer(device=0, k_hint=20) as p:
        p.set_param("wave_force", 1)
        p.build(src)
        idx, d2 = p.query(tgt, 20)
        assert p.stats()["n_wave"] > 0
    wi, wd = oracle.knn_bruteforce(src.astype(np.float64), tgt.astype(np.float64), 20)
    assert np.array_equal(idx, wi) and np.array_equal(d2, wd)


@pytest.mark.parametrize("k,mode,tile", [(8, 0, 1), (8, 1, 1), (16, 0, 1), (20, 1, 1), (32, 0, 1), (8, 1, 0)])
def test_fused_query_blend_matches_the_two_calls(pkg, oracle, k, mode, tile):
    """pt_query_blend_resident = pt_query_resident + pt_blend_dev: identical neighbours and distances, blended attributes
    within the path's tolerance of the oracle -- on the tile kernel, its hand-over list, and the group kernel alone (tile = 0)."""
    import torch
    n, m, seed = 300000, 20000, 0xB1
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.set_param("tile", tile)
        p.build_synth(n, seed)
        p.targets_synth(m, seed)
        i0 = torch.empty((m, k), dtype=torch.int32, device="cuda"); d0 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, i0, d0)
        i1 = torch.full((m, k), 7, dtype=torch.int32, device="cuda"); d1 = torch.zeros((m, k), dtype=torch.float64, device="cuda")
        rgb = torch.full((m, 3), -1.0, dtype=torch.float32, device="cuda"); nrm = torch.full((m, 3), -9.0, dtype=torch.float32, device="cuda")
        p.query_blend_resident_dev(k, mode, i1, d1, rgb, nrm)
        torch.cuda.synchronize()
        left = p.stats()["n_leftover"]
        assert torch.equal(i0, i1) and torch.equal(d0, d1)
        if tile:
            assert 0 < left < m, "the case should exercise both the tile kernel and its hand-over list (left=%d)" % left
    rc, rn = oracle.blend(i0.cpu().numpy().view(np.uint32), d0.cpu().numpy(), oracle.synth_rgb(seed, n), oracle.synth_nrm(seed, n), mode)
    assert np.abs(rgb.cpu().numpy() - rc).max() / 255 <= TOL and np.abs(nrm.cpu().numpy() - rn).max() <= TOL


@pytest.mark.parametrize("k,mode,tile,thr,f64", [(8, 0, 0, 0, False), (20, 1, 0, 16, False), (32, 0, 1, 0, False), (8, 1, 1, 24, True)])
def test_fused_query_blend_on_the_wave_kernel(pkg, oracle, k, mode, tile, thr, f64):
    """Clouds whose targets go to the one-wave-per-target kernel (forced here; with and without refined cells; alone or behind the
    tile kernel): the blend happens inside that launch, one gather per lane -- same neighbours as the plain query, blended
    attributes within tolerance of the oracle, for every target."""
    import torch
    n, m, seed = 200000, 8000, 0xB7
    dist = pkg.capi.DIST_CLUSTERED
    xt = pkg.F64 if f64 else pkg.F32
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.set_param("tile", tile); p.set_param("wave_force", 1); p.set_param("refine_threshold", thr)
        p.build_synth(n, seed, dist=dist, xyz_type=xt)
        p.targets_synth(m, seed, dist=dist, xyz_type=xt)
        assert (p.stats()["n_nodes"] > 0) == (thr > 0)
        i0 = torch.empty((m, k), dtype=torch.int32, device="cuda"); d0 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, i0, d0)
        i1 = torch.full((m, k), 7, dtype=torch.int32, device="cuda"); d1 = torch.zeros((m, k), dtype=torch.float64, device="cuda")
        rgb = torch.full((m, 3), -1.0, dtype=torch.float32, device="cuda"); nrm = torch.full((m, 3), -9.0, dtype=torch.float32, device="cuda")
        p.query_blend_resident_dev(k, mode, i1, d1, rgb, nrm)
        torch.cuda.synchronize()
        assert p.stats()["n_wave"] > 0
        assert torch.equal(i0, i1) and torch.equal(d0, d1)
    rc, rn = oracle.blend(i0.cpu().numpy().view(np.uint32), d0.cpu().numpy(), oracle.synth_rgb(seed, n), oracle.synth_nrm(seed, n), mode)
    assert np.abs(rgb.cpu().numpy() - rc).max() / 255 <= TOL and np.abs(nrm.cpu().numpy() - rn).max() <= TOL


def test_fused_query_blend_double_cloud_and_errors(pkg):
    """fp64 clouds behind the fused call: same neighbours as the two calls, blend within tolerance; inverse-d2 without a
    d2 buffer is refused."""
    import torch
    n, m, k, seed = 40000, 3000, 8, 0xD0
    with pkg.PointsTransfer(device=0) as p:
        p.build_synth(n, seed, xyz_type=pkg.F64)
        p.targets_synth(m, seed, xyz_type=pkg.F64)
        i0 = torch.empty((m, k), dtype=torch.int32, device="cuda"); d0 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        r0 = torch.empty((m, 3), dtype=torch.float32, device="cuda"); n0 = torch.empty((m, 3), dtype=torch.float32, device="cuda")
        p.query_resident_dev(k, i0, d0)
        p.blend_dev(i0, d0, m, k, pkg.BLEND_INV_D2, r0, n0)
        i1 = torch.empty_like(i0); d1 = torch.empty_like(d0); r1 = torch.empty_like(r0); n1 = torch.empty_like(n0)
        p.query_blend_resident_dev(k, pkg.BLEND_INV_D2, i1, d1, r1, n1)
        torch.cuda.synchronize()
        assert torch.equal(i0, i1) and torch.equal(d0, d1)
        # (fp64 clouds run the tile kernel too: the fused blend sums the same terms in another order)
        assert float((r0 - r1).abs().max()) / 255 <= TOL and float((n0 - n1).abs().max()) <= TOL
        with pytest.raises(Exception):
            p.query_blend_resident_dev(k, pkg.BLEND_INV_D2, i1, None, r1, n1)


def test_sampled_bounding_box_and_its_fallback(pkg, oracle):
    """Big clouds (>= "guess_min_points", lowered here) lay their grid out from a sampled bounding box that pass 1
    verifies.  An outlier the sample misses must send the build back to the exact box -- same neighbours either way."""
    n, m, k, seed = 400_000, 3000, 8, 0x5A
    src = oracle.synth_xyz(seed, 0, n)
    tgt = oracle.synth_xyz(seed, 1, m)
    with pkg.PointsTransfer(device=0, rho=0.5) as p:          # rho 0.5: enough blocks for the two-level (chunked) sort
        p.set_param("guess_min_points", 100000)
        p.set_param("grid_hint", 0)                            # (at rho 0.5 even a uniform cloud gets its cell size refined; a rebuild that starts from the
                                                               #  remembered size lays out a grid without room for the guess -- not what this test is about)
        p.build(src)
        assert p.stats()["bbox_guess"] == 1 and p.stats()["n_levels"] == 2
        _check_exact(p.query(tgt, k), oracle.KdTree(src).query(tgt, k), "sampled box accepted")
        p.rebuild()
        assert p.stats()["bbox_guess"] == 1
    out = src.copy()
    out[:, 12345] = (3.0, -2.0, 7.5)                        # not in one of the sampled runs
    tgt2 = np.concatenate([tgt, np.array([[2.9, 3.0], [-1.9, -2.0], [7.4, 7.5]], np.float32)], axis=1)
    want = oracle.KdTree(out).query(tgt2, k)
    with pkg.PointsTransfer(device=0, rho=0.5) as p:
        p.set_param("guess_min_points", 100000)
        p.build(out)
        assert p.stats()["bbox_guess"] == -1
        _check_exact(p.query(tgt2, k), want, "sampled box rejected")
        p.rebuild()
        assert p.stats()["bbox_guess"] == 0                  # no second guess for this cloud
        _check_exact(p.query(tgt2, k), want, "exact box")


def test_weighted_blend_is_the_reference_mix_bit_for_bit(pkg, oracle):
    """pt_blend_weighted against the oracle's restatement of pointsTransfer.cpp:95-97: identical floats (k = 3 barycentric
    weights as in the reference, and longer lists with missing entries)."""
    rng = np.random.default_rng(9)
    n = 5000
    src = rng.random((3, n), dtype=np.float32)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8); nrm = rng.standard_normal((n, 3)).astype(np.float32)
    with pkg.PointsTransfer(device=0) as p:
        p.build(src, rgb=rgb, nrm=nrm)
        for m, k in ((4000, 3), (1000, 20), (1, 1)):
            idx = rng.integers(0, n, size=(m, k)).astype(np.uint32)
            idx[rng.random((m, k)) < 0.05] = 0xFFFFFFFF
            w = rng.random((m, k))
            if k == 3:
                w /= w.sum(axis=1, keepdims=True)                                 # barycentric
            gr, gn = p.blend_weighted(idx, w)
            wr, wn = oracle.blend_weighted(idx, w, rgb, nrm)
            assert np.array_equal(gr, wr) and np.array_equal(gn, wn), (m, k)


def test_caller_targets_made_resident(pkg, oracle):
    """pt_targets_soa / pt_targets_aos: the caller's own targets behind pt_query_resident and the fused call -- same answers
    as the transient query, host and device arrays, planar fp32 and AoS Point records."""
    import torch
    n, m, k, seed = 80000, 5000, 8, 0x7A
    src = oracle.synth_xyz(seed, 0, n); tgt = oracle.synth_xyz(seed, 1, m)
    rgbs = oracle.synth_rgb(seed, n); nrms = oracle.synth_nrm(seed, n)
    want = oracle.KdTree(src).query(tgt, k)
    wr, wn = oracle.blend(want[0], want[1], rgbs, nrms, 0)
    with pkg.PointsTransfer(device=0) as p:
        p.build(src, rgb=rgbs, nrm=nrms)
        for dev_side in (False, True):
            p.set_targets(torch.from_numpy(tgt).cuda() if dev_side else tgt, pkg.F32)
            assert p.num_targets == m
            idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
            rgb = torch.empty((m, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((m, 3), dtype=torch.float32, device="cuda")
            p.query_blend_resident_dev(k, pkg.BLEND_MEAN, idx, d2, rgb, nrm)
            torch.cuda.synchronize()
            _check_exact((idx.cpu().numpy().view(np.uint32), d2.cpu().numpy()), want, "resident caller targets")
            assert np.abs(rgb.cpu().numpy() - wr).max() / 255 <= TOL and np.abs(nrm.cpu().numpy() - wn).max() <= TOL
    pts = np.zeros(n, dtype=pkg.POINT_DTYPE); pts["ver"] = src.T.astype(np.float64); pts["color"] = rgbs; pts["normal"] = nrms
    tp = np.zeros(m, dtype=pkg.POINT_DTYPE); tp["ver"] = tgt.T.astype(np.float64)
    with pkg.PointsTransfer(device=0) as p:
        p.build_aos(pts)
        p.set_targets_aos(tp)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        _check_exact((idx.cpu().numpy().view(np.uint32), d2.cpu().numpy()), want, "resident AoS targets")


def test_double_cloud_with_coordinates_beyond_float_range(pkg, oracle):
    """fp64 clouds whose coordinates an fp32 cannot hold (the tile kernel's fp32 shadow would be inf) still get exact answers."""
    rng = np.random.default_rng(12)
    src = rng.random((3, 20000)) * 1e39
    tgt = rng.random((3, 500)) * 1e39
    with pkg.PointsTransfer(device=0) as p:
        p.build(src, xyz_type=pkg.F64)
        got = p.query(tgt, 8, xyz_type=pkg.F64)
    _check_exact(got, oracle.knn_bruteforce(src, tgt, 8), "huge coordinates")


# ---- texture bake: atlas bytes against the oracle ----------------------------------------------------
@pytest.mark.parametrize("seed,n,grid,k,R,degenerate,f64", [(1, 6000, 6, 20, 512, False, True), (2, 20000, 9, 20, 700, False, False),
                                                           (5, 3000, 3, 8, 128, True, True), (7, 1500, 2, 32, 257, True, False),
                                                           (3, 5000, 5, 20, 300, False, "f16")])
def test_texture_bake_matches_oracle(pkg, oracle, seed, n, grid, k, R, degenerate, f64):
    from _bake_cases import make_case, point_records
    src, rgb, verts, uv, vrgb, faces = make_case(seed, n=n, grid=grid, degenerate=degenerate)
    half = f64 == "f16"                     # fp16-resident cloud: the bake reads its fp32 image
    f64 = f64 is True
    if half:
        src = src.astype(np.float16).astype(np.float64)
    if not f64:
        src = src.astype(np.float32).astype(np.float64); verts = verts.astype(np.float32).astype(np.float64)
    if degenerate:
        uv = uv * 1.3 - 0.15
        faces = np.vstack([faces, [[0, 1, 99999]]]).astype(np.int32)
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        if f64:
            p.build_aos(point_records(pkg.POINT_DTYPE, src, rgb))
            vrec = point_records(pkg.POINT_DTYPE, verts, vrgb, uv)
            idx, d2 = p.query_aos(vrec, k)
        else:
            p.build(src.astype(np.float16 if half else np.float32), rgb, np.zeros((n, 3), np.float32))
            vrec = point_records(pkg.POINT_DTYPE, verts, vrgb, uv)
            idx, d2 = p.query(verts.astype(np.float32), k)
        wi, wd = oracle.knn_bruteforce(src, verts, k)
        assert np.array_equal(idx, wi)
        if degenerate:
            idx = idx.copy(); idx[3, :5] = 0xFFFFFFFF
        got = p.bake_texture(vrec, faces, idx, R)
        want = oracle.bake_texture(src, rgb, verts, uv, vrgb, faces, idx, R)
        assert (want[:, :, 3] == 255).mean() > 0.3
        assert np.array_equal(got, want)
        # edge padding: alone, and fused into the bake call
        wpad = oracle.dilate_pad(want, 25)
        assert np.array_equal(p.texture_pad(got, 25), wpad)
        assert np.array_equal(p.bake_texture(vrec, faces, idx, R, pad_ksize=25), wpad)
        assert p.stats()["ms_bake"] > 0


def test_texture_pad_matches_oracle_on_noise(pkg, oracle):
    rng = np.random.default_rng(8)
    R = 333
    tex = np.zeros((R, R, 4), np.uint8)
    m = rng.random((R, R)) < 0.05
    tex[m] = rng.integers(0, 256, size=(int(m.sum()), 4), dtype=np.uint8)
    with pkg.PointsTransfer(device=0) as p:
        for ks in (1, 5, 25):
            assert np.array_equal(p.texture_pad(tex, ks), oracle.dilate_pad(tex, ks))
        with pytest.raises(pkg.PtError):
            p.texture_pad(tex, 4)


def _read_png_rgba(path):
    """minimal PNG reader for the files host/png_write.h produces (8-bit RGBA, filter type 0 on every row)"""
    import struct, zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    off, idat, w, h = 8, [], 0, 0
    while off < len(data):
        ln, typ = struct.unpack(">I4s", data[off:off + 8])
        body = data[off + 8:off + 8 + ln]
        assert struct.unpack(">I", data[off + 8 + ln:off + 12 + ln])[0] == (zlib.crc32(typ + body) & 0xFFFFFFFF), "chunk CRC"
        if typ == b"IHDR":
            w, h, depth, ctype, comp, flt, inter = struct.unpack(">IIBBBBB", body)
            assert (depth, ctype, comp, flt, inter) == (8, 6, 0, 0, 0)
        elif typ == b"IDAT":
            idat.append(body)
        off += 12 + ln
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, w * 4 + 1)      # (checks the Adler-32 of the whole stream)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 4)


def test_cli_reads_cloud_fields_by_name(tmp_path, pkg):
    """A cloud file whose vertex element declares the nine fields under their usual names in ANOTHER order, with other properties beside
    them (what scanner exports look like): the CLI takes the fields by name (host/ply_fast.h cloud_layout) and finds the neighbours it finds
    in the plain positional file of the same points -- single process and through the --gpus 1 rank path."""
    rng = np.random.default_rng(8)
    n, m = 4000, 200
    cloud = np.round(rng.random((n, 3)) * [3, 2, 1], 5); cn = np.round(rng.standard_normal((n, 3)), 4); crgb = rng.integers(0, 256, (n, 3))
    verts = np.round(rng.random((m, 3)) * [3, 2, 1], 5)
    hdr = "ply\nformat ascii 1.0\nelement vertex %d\n" % n
    with open(tmp_path / "pos.ply", "w") as f:
        f.write(hdr + "".join("property float %s\n" % p for p in ("x", "y", "z", "nx", "ny", "nz")) + "".join("property uchar %s\n" % p for p in ("red", "green", "blue")) + "end_header\n")
        for p_, q_, c_ in zip(cloud, cn, crgb):
            f.write("%.5f %.5f %.5f %.4f %.4f %.4f %d %d %d\n" % (*p_, *q_, *c_))
    with open(tmp_path / "named.ply", "w") as f:
        f.write(hdr + "property float scalar_intensity\nproperty uchar blue\nproperty float nz\nproperty double y\nproperty uchar red\nproperty double x\nproperty float ny\n"
                "property uchar alpha\nproperty double z\nproperty float nx\nproperty uchar green\nend_header\n")
        for p_, q_, c_ in zip(cloud, cn, crgb):
            f.write("0.5 %d %.4f %.5f %d %.5f %.4f 255 %.5f %.4f %d\n" % (c_[2], q_[2], p_[1], c_[0], p_[0], q_[1], p_[2], q_[0], c_[1]))
    with open(tmp_path / "mesh.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nelement face 1\nend_header\n" % m)
        for v in verts:
            f.write("%.5f %.5f %.5f 0 0 1 0.5 0.5 1 2 3\n" % tuple(v))
        f.write("3 0 1 2\n")
    exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "pointsTransfer")
    outs = {}
    for name, extra in (("pos", []), ("named", []), ("named_sharded", ["--gpus", "1"])):
        d = tmp_path / name; d.mkdir()
        r = subprocess.run([exe, str(tmp_path / ("pos.ply" if name == "pos" else "named.ply")), str(tmp_path / "mesh.ply"), "--resolution", "64", "--k", "8"] + extra,
                           capture_output=True, text=True, cwd=d, timeout=300)
        assert r.returncode == 0, r.stderr
        outs[name] = open(d / "transfer.ply").read()
    assert outs["pos"] == outs["named"]
    rows = lambda t: np.array([[float(v) for v in l.split()] for l in t.split("end_header\n")[1].strip().splitlines()[:m]])
    a, b = rows(outs["pos"]), rows(outs["named_sharded"])
    assert np.abs(a[:, 8:] - b[:, 8:]).max() <= 1 and np.abs(a[:, :8] - b[:, :8]).max() <= 2e-5


def _write_binary_plys(pc, mesh, src, rgb, verts, uv, vrgb, faces):
    """binary little-endian PLY files: cloud (double xyz, float normals, uchar colours), mesh (double x y z nx ny nz s t, int colours,
    list uchar int faces)"""
    n, m = src.shape[1], verts.shape[1]
    cd = np.dtype([("p", "<f8", 3), ("n", "<f4", 3), ("c", "u1", 3)])
    a = np.zeros(n, cd); a["p"] = src.T; a["n"] = (0, 0, 1); a["c"] = rgb
    with open(pc, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\nproperty double z\n"
                 "property float nx\nproperty float ny\nproperty float nz\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n).encode())
        f.write(a.tobytes())
    md = np.dtype([("p", "<f8", 3), ("n", "<f8", 3), ("uv", "<f8", 2), ("c", "<i4", 3)])
    b = np.zeros(m, md); b["p"] = verts.T; b["n"] = (0, 0, 1); b["uv"] = uv; b["c"] = vrgb
    fd = np.dtype([("k", "u1"), ("v", "<i4", 3)])
    fc = np.zeros(len(faces), fd); fc["k"] = 3; fc["v"] = faces
    with open(mesh, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\nproperty double z\n"
                 "property double nx\nproperty double ny\nproperty double nz\nproperty double s\nproperty double t\nproperty int red\n"
                 "property int green\nproperty int blue\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n" % (m, len(faces))).encode())
        f.write(b.tobytes()); f.write(fc.tobytes())


def _write_ascii_plys(pc, mesh, src, rgb, verts, uv, vrgb, faces):
    n, m = src.shape[1], verts.shape[1]
    with open(pc, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\n"
                "property float ny\nproperty float nz\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n)
        for i in range(n):
            f.write("%.17g %.17g %.17g 0 0 1 %d %d %d\n" % (*src[:, i], *rgb[i]))
    with open(mesh, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\n"
                "property float ny\nproperty float nz\nproperty float s\nproperty float t\nproperty uchar red\nproperty uchar green\n"
                "property uchar blue\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n" % (m, len(faces)))
        for i in range(m):
            f.write("%.17g %.17g %.17g 0 0 1 %.17g %.17g %d %d %d\n" % (*verts[:, i], *uv[i], *vrgb[i]))
        for fc in faces:
            f.write("3 %d %d %d\n" % tuple(fc))


def test_cli_synthetic_runs_the_baseline_configs_without_files(tmp_path, pkg, oracle):
    """`pointsTransfer - - --synthetic N M SEED`: the generator of Appendix C instead of the two files, the reference's report
    lines in the reference's order, `--neighbors` the index matrix -- config 1 (10k / 1k / k = 1, seed 0xC1) against the oracle's brute force."""
    exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "pointsTransfer")
    n, m, k, seed = 10_000, 1_000, 1, 0xC1
    nb = tmp_path / "nbr.bin"; js = tmp_path / "s.json"
    r = subprocess.run([exe, "-", "-", "--synthetic", str(n), str(m), hex(seed), "--k", str(k), "--neighbors", str(nb), "--json", str(js)], capture_output=True, text=True,
                       cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    heads = ["PC Point count: %d" % n, "Read point set in:", "Built Kd tree in:", "Mesh vertex count: %d" % m, "Mesh face count: 0", "Read mesh faces:",
             "Neighbor search total time:", "Draw triangles total time:", "Output time:", "Total real time:", "VIRT:", "RES:  "]
    assert len(lines) == len(heads) and all(l.startswith(h) for l, h in zip(lines, heads)), r.stdout
    got = np.fromfile(nb, np.uint32).reshape(m, k)
    wi, _ = oracle.knn_bruteforce(oracle.synth_xyz(seed, 0, n), oracle.synth_xyz(seed, 1, m), k)
    assert np.array_equal(got, wi)
    import json as _json
    assert _json.load(open(js))["synthetic"]["n"] == n
    # config 2's shape (10M / 1M / k = 8) through the same binary: properties only (sorted distances are not written; every index valid)
    r = subprocess.run([exe, "-", "-", "--synthetic", "10000000", "1000000", "0xC2", "--k", "8", "--neighbors", str(nb)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(nb, np.uint32).reshape(1_000_000, 8)
    assert got.max() < 10_000_000 and (np.sort(got, axis=1)[:, 1:] != np.sort(got, axis=1)[:, :-1]).all()


@pytest.mark.parametrize("fmt", ["ascii", "binary"])
def test_cli_writes_the_texture(tmp_path, pkg, oracle, fmt):
    """pointsTransfer cloud.ply mesh.ply -> texture.png (reference src/pointsTransfer.cpp:613-615): search (K = 20), per-face bake,
    25 x 25 edge padding and PNG writing, end to end through the C++ host, decoded and compared with the oracle's atlas -- from the
    reference's ASCII files and from binary little-endian ones (the cloud reaches the GPU as planar arrays either way)."""
    import os, subprocess
    from _bake_cases import make_case
    src, rgb, verts, uv, vrgb, faces = make_case(9, n=8000, grid=5)
    pc, mesh = tmp_path / "cloud.ply", tmp_path / "mesh.ply"
    (_write_ascii_plys if fmt == "ascii" else _write_binary_plys)(pc, mesh, src, rgb, verts, uv, vrgb, faces)
    exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "pointsTransfer")
    R = 640
    r = subprocess.run([exe, str(pc), str(mesh), "--resolution", str(R), "--out", ""], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert not os.path.exists(tmp_path / "transfer.ply")
    got = _read_png_rgba(tmp_path / "texture.png")
    idx, _ = oracle.knn_bruteforce(src, verts, 20)
    want = oracle.dilate_pad(oracle.bake_texture(src, rgb, verts, uv, vrgb, faces, idx, R), 25)
    assert got.shape == (R, R, 4)
    assert np.array_equal(got[:, :, [2, 1, 0, 3]], want)                  # the file holds R, G, B, A; the atlas is B, G, R, A


def test_streamed_planar_upload_equals_build_soa(pkg, oracle):
    """pt_upload_begin / pt_upload_range / pt_upload_end (the CLI's ingest: page-locked planar arrays, ranges uploaded as the parser
    finishes them) builds the same cloud as pt_build_soa: same neighbours, same blended attributes."""
    import ctypes as C
    L = pkg.capi.lib()
    rng = np.random.default_rng(21)
    n, m, k = 70000, 3000, 8
    src = rng.random((3, n)); rgb = rng.integers(0, 256, (n, 3), dtype=np.uint8); nrm = rng.standard_normal((n, 3)).astype(np.float32)
    tgt = rng.random((3, m))
    with pkg.PointsTransfer(device=0) as a, pkg.PointsTransfer(device=0) as b:
        a.build(src, rgb, nrm)
        wi, wd = a.query(tgt, k); wc, wn = a.blend(wi, wd)
        sizes = [n * 8, n * 8, n * 8, n * 3, n * 12]
        ptrs = [L.pt_host_alloc(sz) for sz in sizes]
        assert all(ptrs)
        for q, arr in zip(ptrs, [src[0], src[1], src[2], rgb, nrm]):
            C.memmove(q, np.ascontiguousarray(arr).ctypes.data, arr.nbytes)
        assert L.pt_upload_begin(b._h, n, pkg.F64, 1) == 0
        cuts = [0, 1, 999, 1000, 31337, n]
        for f0, f1 in reversed(list(zip(cuts[:-1], cuts[1:]))):             # ranges in any order
            assert L.pt_upload_range(b._h, f0, f1 - f0, ptrs[0] + 8 * f0, ptrs[1] + 8 * f0, ptrs[2] + 8 * f0, ptrs[3] + 3 * f0, ptrs[4] + 12 * f0) == 0
        assert L.pt_upload_range(b._h, n - 1, 5, ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4]) == pkg.capi.ERR_ARG      # beyond the cloud
        assert L.pt_upload_end(b._h) == 0
        for q in ptrs:
            L.pt_host_free(q)
        gi, gd = b.query(tgt, k); gc, gn = b.blend(gi, gd)
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd) and np.array_equal(gc, wc) and np.array_equal(gn, wn)
        assert L.pt_upload_end(b._h) == pkg.capi.ERR_STATE                   # no upload in progress
    ri, rd = oracle.knn_bruteforce(src, tgt, k)
    assert np.array_equal(wi, ri) and np.array_equal(wd, rd)


@pytest.mark.parametrize("dtype,nchunks", [(np.float32, 2), (np.float64, 3), (np.float32, 7)])
def test_streamed_source_equals_resident(pkg, oracle, dtype, nchunks):
    """a cloud kept in host memory and streamed through the GPU in chunks (each gridded and searched like a
    resident cloud, the running k best merged under (d2, index)) gives the resident search's result bit for bit, with 64-bit ids."""
    rng = np.random.default_rng(31)
    n, m, k = 120_001, 4000, 20
    src = rng.random((3, n)).astype(dtype)
    src[:, 5000:5200] = src[:, :200]                                       # duplicates across chunk borders: ties broken by the global index
    tgt = rng.random((3, m)).astype(dtype)
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build(src)
        wi, wd = p.query(tgt, k)
        p.set_targets(tgt)
        chunk = (n + nchunks - 1) // nchunks
        gi, gd = p.stream_query(src, chunk, k)
        assert np.array_equal(gi, wi.astype(np.uint64)) and np.array_equal(gd, wd)
        base = (1 << 33) + 12345                                            # ids beyond 32 bits
        gi2, gd2 = p.stream_query(src, chunk, k, first_id=base)
        assert np.array_equal(gi2, wi.astype(np.uint64) + np.uint64(base)) and np.array_equal(gd2, wd)
        with pytest.raises(pkg.PtError):                                     # the chunks are gone: nothing resident to query
            p.query(tgt, k)
        # k larger than a chunk, an empty cloud
        small = src[:, :30]
        p.build(small); si, sd = p.query(tgt[:, :50], 32); p.set_targets(tgt[:, :50])
        ti, td = p.stream_query(small, 7, 32)
        assert np.array_equal(ti[:, :30], si[:, :30].astype(np.uint64)) and (ti[:, 30:] == np.uint64(0xFFFFFFFFFFFFFFFF)).all() and np.isinf(td[:, 30:]).all()
        ei, ed = p.stream_query(src[:, :0], 1000, 8)
        assert (ei == np.uint64(0xFFFFFFFFFFFFFFFF)).all() and np.isinf(ed).all()
    ri, rd = oracle.knn_bruteforce(src.astype(np.float64), tgt.astype(np.float64), k)
    assert np.array_equal(wi, ri) and np.array_equal(wd, rd)


@pytest.mark.parametrize("dtype,k", [(np.float32, 8), (np.float32, 20), (np.float64, 8)])
def test_streamed_source_in_spatial_order_skips_chunks(pkg, oracle, dtype, k):
    """A cloud stored in spatial order (sorted along x, as scanners and tiled exports deliver it): from the second chunk on the
    targets' current k-th distances bound the search, chunks no target can reach are skipped (pt_stats.stream_skipped), and the
    answer stays the resident search's, bit for bit -- targets far outside the cloud and in its gaps included."""
    rng = np.random.default_rng(57)
    n, m = 400_000, 6000
    src = rng.random((3, n))
    src[0] = np.sort(src[0] * 8.0)                                          # eight unit cubes side by side, in x order
    src[:, 100_000:100_300] = src[:, 99_700:100_000]                        # duplicates across a chunk border
    src = src.astype(dtype)
    tgt = rng.random((3, m)); tgt[0] = tgt[0] * 0.9                         # all targets near the first chunks ...
    tgt[:, :50] += 20.0                                                     # ... except a few far outside the cloud (they reach everything)
    tgt = tgt.astype(dtype)
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build(src)
        wi, wd = p.query(tgt, k)
        far = np.zeros(m, bool); far[:50] = True
        p.set_targets(tgt[:, ~far])
        gi, gd = p.stream_query(src, n // 8, k)
        assert p.stats()["stream_skipped"] >= 5                              # chunks 2 .. 7 lie beyond every near target's k-th distance
        assert np.array_equal(gi, wi[~far].astype(np.uint64)) and np.array_equal(gd, wd[~far])
        assert p.stats()["stream_revisited"] == 0                            # every target lay inside the first chunk's box: nothing deferred
        p.set_targets(tgt)                                                   # the far targets lie outside EVERY chunk: deferred by the forward sweep,
        gi, gd = p.stream_query(src, n // 8, k)                              # picked up -- unbounded, from outside -- by the backward sweep's first chunk
        st = p.stats()
        assert st["stream_skipped"] >= 5 and 1 <= st["stream_revisited"] <= 2, st
        assert np.array_equal(gi, wi.astype(np.uint64)) and np.array_equal(gd, wd)
        # targets spread over the whole cloud: each is searched first in its own slab's chunk, the ones near a slab's lower border bring
        # the chunk before it back
        wide = tgt[:, 50:].copy(); wide[0] = (rng.random(m - 50) * 8.0).astype(dtype)     # (without the far ones: those reach every slab alike)
        p.build(src); ui, ud = p.query(wide, k)
        p.set_targets(wide)
        gi, gd = p.stream_query(src, n // 8, k)
        st = p.stats()
        assert np.array_equal(gi, ui.astype(np.uint64)) and np.array_equal(gd, ud)
        assert st["stream_revisited"] <= 2, st                                # (targets within a few point spacings of a slab were served by it in the forward sweep)
    ri, rd = oracle.knn_bruteforce(src.astype(np.float64)[:, :], tgt.astype(np.float64)[:, 50:250], k)
    assert np.array_equal(wi[50:250], ri) and np.array_equal(wd[50:250], rd)


def test_streamed_source_100m_equals_resident(pkg, oracle):
    """the streamed search at 100 M points (four chunks, the bounded tile kernel from the second on) against the
    resident search of the same cloud, bit for bit; a sample of the rows against the oracle's brute force."""
    n, m, k, seed = 100_000_000, 400_000, 8, 0x5E
    src = oracle.synth_xyz(seed, 0, n)
    tgt = oracle.synth_xyz(seed, 1, m)
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build(src)
        wi, wd = p.query(tgt, k)
        p.set_targets(tgt)
        gi, gd = p.stream_query(src, n // 4, k)
    assert np.array_equal(gi, wi.astype(np.uint64)) and np.array_equal(gd, wd)
    rows = np.arange(0, m, m // 40)
    sub = np.abs(src[0][None, :] - tgt[0][rows][:, None]).min(axis=0) < 0.02   # (the brute force only needs the points near the sampled rows in x)
    cand = np.nonzero(sub)[0]
    ri, rd = oracle.knn_bruteforce(src[:, cand], tgt[:, rows], k)
    assert np.array_equal(cand[ri].astype(np.uint32), wi[rows]) and np.array_equal(rd, wd[rows])


# ---- native slab exchange behind the C ABI (pt_exchange_*): same phases as the RCCL path, device copies as transport ------------
@pytest.mark.parametrize("g,k,f64,sharded", [(2, 8, False, False), (3, 20, True, False), (5, 8, False, False), (2, 8, False, True), (3, 20, True, True), (5, 16, False, True),
                                             (8, 8, False, True), (8, 20, True, False)])
def test_native_exchange_on_logical_slabs(pkg, oracle, g, k, f64, sharded):
    """G contexts of one process, one slab each (equal-count quantiles along x; the last slab may hold NO targets): home search,
    then pt_exchange_merge_local -- count matrix, owner-to-owner requests, bounded answers, merge, re-blend of the completed rows.
    Result: the global search bit for bit, and the blend of every row within 1e-5 of the oracle's.
    sharded (round 4): every slab keeps LOCAL ids in its records ("local_ids": ascending global indices) and the attribute records of its
    OWN points only (set_attributes_local); the fused home blend gathers from that table, the answers carry their candidates' records."""
    import torch
    n, m, seed = 150000, 9000, 0xE0 + g
    dt = np.float64 if f64 else np.float32
    xt = pkg.F64 if f64 else pkg.F32
    src = oracle.synth_xyz(seed, 0, n).astype(dt); tgt = oracle.synth_xyz(seed, 1, m).astype(dt)
    tgt = tgt[:, tgt[0] < 0.93]                                              # nothing homed near the far end: an empty last rank at g = 5
    m = tgt.shape[1]
    rgb, nrm = oracle.synth_rgb(seed, n), oracle.synth_nrm(seed, n)
    want_i, want_d = oracle.KdTree(src.astype(np.float64)).query(tgt.astype(np.float64), k)
    bounds = [-math.inf] + [float(v) for v in np.quantile(src[0], np.arange(1, g) / g)] + [math.inf]
    if g == 5:
        bounds[-2] = 0.95
    home = np.clip(np.searchsorted(np.array(bounds), tgt[0], side="right") - 1, 0, g - 1)
    pts, xs, ii, dd, cc, nn, rows = [], [], [], [], [], [], []
    for s in range(g):
        p = pkg.PointsTransfer(device=0, k_hint=k)
        sel = np.nonzero((src[0] >= bounds[s]) & (src[0] < bounds[s + 1]))[0]
        if sharded:
            p.set_param("local_ids", 1)
        p.build(np.ascontiguousarray(src[:, sel]), gidx=sel.astype(np.uint32))
        if sharded:
            p.set_attributes_local(rgb[sel], nrm[sel])                       # this slab's points only: 1 / g of the table
            with pytest.raises(pkg.PtError):
                p.set_attributes(rgb, nrm)
        else:
            p.set_attributes(rgb, nrm)                                       # the table is replicated: indexed by the global index
        mine = np.nonzero(home == s)[0]
        ms = len(mine)
        x = torch.from_numpy(np.ascontiguousarray(tgt[:, mine])).cuda()
        i_ = torch.empty((ms, k), dtype=torch.int32, device="cuda"); d_ = torch.empty((ms, k), dtype=torch.float64, device="cuda")
        c_ = torch.zeros((ms, 3), dtype=torch.float32, device="cuda"); n_ = torch.zeros((ms, 3), dtype=torch.float32, device="cuda")
        if ms:
            p.query_dev(x, xt, ms, k, i_, d_)
            p.blend_dev(i_, d_, ms, k, pkg.BLEND_MEAN, c_, n_)
            if sharded:                                                      # the home lists name home points, by GLOBAL index
                hi = i_.cpu().numpy().view(np.uint32)
                assert np.isin(hi[hi != pkg.NOIDX], sel).all()
        pts.append(p); xs.append(x); ii.append(i_); dd.append(d_); cc.append(c_); nn.append(n_); rows.append(mine)
    assert g != 5 or len(rows[-1]) == 0
    pkg.PointsTransfer.exchange_merge_local(pts, xs, xt, k, 0, bounds, ii, dd, pkg.BLEND_MEAN, cc, nn)
    torch.cuda.synchronize()
    gi = np.empty((m, k), np.uint32); gd = np.empty((m, k)); gc = np.empty((m, 3), np.float32); gn = np.empty((m, 3), np.float32)
    for s in range(g):
        gi[rows[s]] = ii[s].cpu().numpy().view(np.uint32); gd[rows[s]] = dd[s].cpu().numpy()
        gc[rows[s]] = cc[s].cpu().numpy(); gn[rows[s]] = nn[s].cpu().numpy()
    _check_exact((gi, gd), (want_i, want_d), "native exchange g=%d" % g)
    rc, rn = oracle.blend(want_i, want_d, rgb, nrm, mode=0)
    assert np.abs(gc - rc).max() / 255.0 <= TOL and np.abs(gn - rn).max() <= TOL
    for p in pts:
        p.close()


def test_local_id_slab_matches_the_global_id_slab(pkg, oracle):
    """A slab whose records carry positions instead of global indices ("local_ids"): the same lists -- ties among duplicates resolved by the
    global index, which ascending indices make the positional order -- the fused blend from the slab's own attribute records within 1e-5 of
    the replicated table's, on the tile kernel and on the wave kernel; indices that do not ascend are refused."""
    import torch
    rng = np.random.default_rng(55)
    n_all, n, m, k = 300_000, 120_000, 5000, 16
    src_all = rng.random((3, n_all), dtype=np.float32)
    src_all[:, 0:294000:7] = src_all[:, 3:294000:7]                           # exact duplicates: ties decided by the index
    sel = np.sort(rng.choice(n_all, n, replace=False)).astype(np.uint32)
    src = np.ascontiguousarray(src_all[:, sel])
    tgt = rng.random((3, m), dtype=np.float32)
    tgt[:, :500] = src[:, rng.integers(0, n, 500)]
    rgb, nrm = oracle.synth_rgb(9, n_all), oracle.synth_nrm(9, n_all)
    wi, wd = oracle.knn_bruteforce(src, tgt, k, gidx=sel)
    rc, rn = oracle.blend(wi, wd, rgb, nrm, mode=1)
    for tile in (1, 0):
        with pkg.PointsTransfer(device=0, k_hint=k) as p:
            p.set_param("local_ids", 1); p.set_param("tile", tile)
            if tile == 0:
                p.set_param("wave_force", 1)
            p.build(src, gidx=sel)
            p.set_attributes_local(rgb[sel], nrm[sel])
            p.set_targets(tgt)
            idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
            c_ = torch.empty((m, 3), dtype=torch.float32, device="cuda"); n_ = torch.empty((m, 3), dtype=torch.float32, device="cuda")
            p.query_blend_resident_dev(k, pkg.BLEND_INV_D2, idx, d2, c_, n_)
            torch.cuda.synchronize()
            _check_exact((idx.cpu().numpy().view(np.uint32), d2.cpu().numpy()), (wi, wd), "local ids, tile=%d" % tile)
            assert np.abs(c_.cpu().numpy() - rc).max() / 255.0 <= TOL and np.abs(n_.cpu().numpy() - rn).max() <= TOL
            c2 = torch.empty_like(c_); n2 = torch.empty_like(n_)
            p.blend_dev(idx, d2, m, k, pkg.BLEND_INV_D2, c2, n2)              # a finished list (global indices) against the local table
            torch.cuda.synchronize()
            assert np.abs(c2.cpu().numpy() - rc).max() / 255.0 <= TOL and np.abs(n2.cpu().numpy() - rn).max() <= TOL
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.set_param("local_ids", 1)
        bad = sel.copy(); bad[[10, 11]] = bad[[11, 10]]
        with pytest.raises(pkg.PtError):
            p.build(src, gidx=bad)


@pytest.mark.parametrize("clustered", [False, True])
def test_generated_slabs_in_index_order_carry_their_own_attributes(pkg, oracle, clustered):
    """bench.py --gpus N as round 4 builds it: every rank GENERATES its slab in index order ("local_ids" before build_synth: per-workgroup
    counts, a scan, ranked writes), so its records carry positions and its attribute table holds its own n / G records; the native exchange
    (answers carrying their candidates' records) then gives the global search bit for bit and the oracle's blend within 1e-5."""
    import torch
    g, n, m, k, seed = 3, 200_000, 6000, 8, 0xA7
    dist = 1 if clustered else 0
    xt = pkg.F16 if clustered else pkg.F32
    src = oracle.synth_xyz(seed, 0, n, dist=dist, n_total=n, m_total=m); tgt = oracle.synth_xyz(seed, 1, m, dist=dist, n_total=n, m_total=m)
    if clustered:
        src = src.astype(np.float16).astype(np.float32)
    rgb, nrm = oracle.synth_rgb(seed, n), oracle.synth_nrm(seed, n)
    want_i, want_d = oracle.KdTree(src.astype(np.float64)).query(tgt.astype(np.float64), k)
    bounds = [-math.inf] + [float(v) for v in np.quantile(src[0], np.arange(1, g) / g)] + [math.inf]
    home = np.clip(np.searchsorted(np.array(bounds), tgt[0], side="right") - 1, 0, g - 1)
    pts, xs, ii, dd, cc, nn, rows = [], [], [], [], [], [], []
    for s in range(g):
        p = pkg.PointsTransfer(device=0, k_hint=k)
        p.set_param("local_ids", 1)
        p.build_synth(n, seed, xyz_type=xt, dist=dist, slab_axis=0, slab_lo=bounds[s], slab_hi=bounds[s + 1])
        sel = np.nonzero((src[0] >= bounds[s]) & (src[0] < bounds[s + 1]))[0]
        assert p.num_source == len(sel)
        with pytest.raises(pkg.PtError):
            p.set_attributes(rgb, nrm)                                       # the table is the slab's own now
        mine = np.nonzero(home == s)[0]
        ms = len(mine)
        x = torch.from_numpy(np.ascontiguousarray(tgt[:, mine])).cuda()
        i_ = torch.empty((ms, k), dtype=torch.int32, device="cuda"); d_ = torch.empty((ms, k), dtype=torch.float64, device="cuda")
        c_ = torch.zeros((ms, 3), dtype=torch.float32, device="cuda"); n_ = torch.zeros((ms, 3), dtype=torch.float32, device="cuda")
        p.set_targets(np.ascontiguousarray(tgt[:, mine]))
        p.query_blend_resident_dev(k, pkg.BLEND_MEAN, i_, d_, c_, n_)        # the fused home blend, from the local table
        torch.cuda.synchronize()
        hi = i_.cpu().numpy().view(np.uint32)
        assert np.isin(hi[hi != pkg.NOIDX], sel).all()                       # global indices, of home points
        sub_i, sub_d = oracle.knn_bruteforce(src[:, sel], tgt[:, mine[:200]], k, gidx=sel.astype(np.uint32))
        assert np.array_equal(hi[:200], sub_i) and np.array_equal(d_.cpu().numpy()[:200], sub_d)
        pts.append(p); xs.append(x); ii.append(i_); dd.append(d_); cc.append(c_); nn.append(n_); rows.append(mine)
    assert sum(p.num_source for p in pts) == n
    pkg.PointsTransfer.exchange_merge_local(pts, xs, pkg.F32, k, 0, bounds, ii, dd, pkg.BLEND_MEAN, cc, nn)
    torch.cuda.synchronize()
    gi = np.empty((m, k), np.uint32); gd = np.empty((m, k)); gc = np.empty((m, 3), np.float32); gn = np.empty((m, 3), np.float32)
    for s in range(g):
        gi[rows[s]] = ii[s].cpu().numpy().view(np.uint32); gd[rows[s]] = dd[s].cpu().numpy()
        gc[rows[s]] = cc[s].cpu().numpy(); gn[rows[s]] = nn[s].cpu().numpy()
    _check_exact((gi, gd), (want_i, want_d), "generated local-id slabs")
    rc, rn = oracle.blend(want_i, want_d, rgb, nrm, mode=0)
    assert np.abs(gc - rc).max() / 255.0 <= TOL and np.abs(gn - rn).max() <= TOL
    for p in pts:
        p.close()


def test_generated_slab_at_c4_scale_local_ids_equal_global_ids(pkg, oracle):
    """What one rank of `bench.py --gpus 8` builds at BASELINE config 4: its eighth of the 1e9-point cloud, GENERATED in index order with
    positions in its records and its own 125 M attribute records (3.9 M workgroup counts through the scan).  The same slab generated the
    round-3 way (global indices in the records, the whole 16-GB table) must give the same lists bit for bit and the same fused blend; the
    indices are global ones of slab points at the distances reported (regenerated through the oracle)."""
    import torch
    n_total, m, k, seed = 1_000_000_000, 200_000, 8, 0xC4
    lo, hi = 0.375, 0.5
    with pkg.PointsTransfer(device=0, k_hint=k) as p:                            # the slab's targets once (a generated slab's ORDER is whatever the atomics made it)
        p.targets_synth(m * 8, seed, slab_axis=0, slab_lo=lo, slab_hi=hi)
        mm = p.num_targets
        tx = torch.empty((3, mm), dtype=torch.float32, device="cuda")
        p.resident_target_xyz_dev(tx)
    tx = tx.cpu().numpy()
    res = []
    for local in (1, 0):
        with pkg.PointsTransfer(device=0, k_hint=k) as p:
            p.set_param("local_ids", local)
            p.build_synth(n_total, seed, slab_axis=0, slab_lo=lo, slab_hi=hi)
            n = p.num_source
            p.set_targets(tx)
            idx = torch.empty((mm, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((mm, k), dtype=torch.float64, device="cuda")
            c_ = torch.empty((mm, 3), dtype=torch.float32, device="cuda"); n_ = torch.empty((mm, 3), dtype=torch.float32, device="cuda")
            p.query_blend_resident_dev(k, pkg.BLEND_MEAN, idx, d2, c_, n_)
            torch.cuda.synchronize()
            res.append((n, idx.cpu().numpy().view(np.uint32), d2.cpu().numpy(), c_.cpu().numpy(), n_.cpu().numpy()))
    a, b = res
    assert a[0] == b[0] and abs(a[0] - n_total / 8) < 1e-3 * n_total and mm > 0
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.abs(a[3] - b[3]).max() / 255.0 <= TOL and np.abs(a[4] - b[4]).max() <= TOL      # (one table gathered by position, the other by index: the same records)
    for r in np.arange(0, mm, max(1, mm // 12))[:12]:
        for j in range(k):
            pt = oracle.synth_xyz(seed, 0, 1, i0=int(a[1][r, j]))[:, 0].astype(np.float64)
            assert lo <= pt[0] < hi
            d = tx[:, r].astype(np.float64) - pt
            assert (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2] == a[2][r, j]


@pytest.mark.parametrize("sharded,work", [(1, "C4"), (0, "C4"), (1, "C5")])
def test_config4_as_eight_logical_slabs_equals_the_single_context_run(sharded, work):
    """BASELINE config 4 at FULL size the way `bench.py --gpus 8` computes it -- eight slabs of the 1e9-point cloud generated, built, searched
    and exchanged (pt_exchange_merge_local: the RCCL path with device copies as transport), with sharded and with replicated attributes --
    against the single-context run of the whole cloud: all 50 M rows, indices and distances bit for bit, blends within 1e-5
    (tools/rehearse_slabs_c4.py; it also prints the per-slab times DESIGN.md section 7's table quotes).  C5: the same for BASELINE config 5 --
    the clustered fp16 cloud at k = 32, equal-count slabs from a sample's quantiles; this case found the clustered TARGET generator being handed
    the slab's point count instead of the cloud's when the slab keeps local ids)."""
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "rehearse_slabs_c4.py")
    r = subprocess.run([sys.executable, tool, "8", "1e9", str(sharded), work], capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-2000:]
    assert "rows that differ from the single-context run: 0 of 50000000" in r.stdout


def test_rccl_communicator_world_of_one(pkg, oracle):
    """librccl is loaded and a communicator of ONE rank comes up on this GPU (what the multi-GPU path runs on every rank before
    its first exchange); with a world of one the exchange is a no-op that leaves the lists untouched."""
    import torch
    src = oracle.synth_xyz(5, 0, 30000); tgt = oracle.synth_xyz(5, 1, 500)
    with pkg.PointsTransfer(device=0) as p:
        uid = p.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        p.comm_init(1, 0, uid)
        with pytest.raises(pkg.PtError):
            p.comm_init(1, 0, uid)                                            # already initialised
        p.build(src)
        x = torch.from_numpy(tgt).cuda()
        i_ = torch.empty((500, 8), dtype=torch.int32, device="cuda"); d_ = torch.empty((500, 8), dtype=torch.float64, device="cuda")
        p.query_dev(x, pkg.F32, 500, 8, i_, d_)
        before = (i_.clone(), d_.clone())
        st = p.exchange_merge_dev(x, pkg.F32, 500, 8, 0, [-math.inf, math.inf], i_, d_)
        torch.cuda.synchronize()
        assert st["crossing"] == 0 and torch.equal(i_, before[0]) and torch.equal(d_, before[1])
        p.comm_destroy()


def test_native_exchange_two_ranks_over_rccl(pkg, oracle, tmp_path):
    """World size 2 over RCCL proper: needs two visible GPUs (one rank per GPU; RCCL refuses two ranks on one device)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs: this box shows %d (RCCL cannot place two ranks on one device)" % torch.cuda.device_count())
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_worker.py")
    import socket
    with socket.socket() as so:                              # a free rendezvous port on the loop-back interface (as bench.py's launcher picks one)
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), worker, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def test_attribute_table_filled_in_ranges(pkg, oracle):
    """pt_set_attributes_range (what the ranks of `pointsTransfer --gpus N` feed from the pieces of the cloud): the table filled in
    three uneven ranges blends exactly like the table uploaded whole; a range outside the table is refused."""
    n, m, k = 30000, 700, 8
    src, tgt = oracle.synth_xyz(21, 0, n), oracle.synth_xyz(21, 1, m)
    rgb, nrm = oracle.synth_rgb(21, n), oracle.synth_nrm(21, n)
    with pkg.PointsTransfer(device=0) as a, pkg.PointsTransfer(device=0) as b:
        a.build(src, rgb, nrm)
        b.build(src)
        for lo, hi in ((0, 1), (1, 17001), (17001, n)):
            b.set_attributes_range(lo, rgb[lo:hi], nrm[lo:hi], n)
        ia, da = a.query(tgt, k)
        ib, db = b.query(tgt, k)
        assert np.array_equal(ia, ib)
        for mode in (pkg.BLEND_MEAN, pkg.BLEND_INV_D2):
            ca, na = a.blend(ia, da, mode=mode)
            cb, nb = b.blend(ib, db, mode=mode)
            assert np.array_equal(ca, cb) and np.array_equal(na, nb)
        with pytest.raises(pkg.PtError):
            b.set_attributes_range(n - 1, rgb[:2], nrm[:2], n)


@pytest.mark.parametrize("gpus,fmt", [(1, "binary"), (1, "ascii"), (2, "binary"), (2, "ascii")])
def test_cli_sharded_path(tmp_path, pkg, oracle, gpus, fmt):
    """pointsTransfer ... --gpus N: launcher -> one rank process per GPU (RCCL communicator, slab build with global indices, home
    search, native exchange) -> finalize (bake on the referenced points only).  texture.png must equal the single-process run's
    byte for byte after decoding; --gpus 1 runs everywhere, --gpus 2 needs two GPUs."""
    import torch
    if gpus > torch.cuda.device_count():
        pytest.skip("needs %d GPUs: this box shows %d" % (gpus, torch.cuda.device_count()))
    from _bake_cases import make_case
    src, rgb, verts, uv, vrgb, faces = make_case(13, n=9000, grid=5)
    pc, mesh = tmp_path / "cloud.ply", tmp_path / "mesh.ply"
    (_write_binary_plys if fmt == "binary" else _write_ascii_plys)(pc, mesh, src, rgb, verts, uv, vrgb, faces)
    exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "pointsTransfer")
    d1, d2_ = tmp_path / "single", tmp_path / "sharded"
    d1.mkdir(); d2_.mkdir()
    r1 = subprocess.run([exe, str(pc), str(mesh), "--resolution", "400"], capture_output=True, text=True, cwd=d1)
    r2 = subprocess.run([exe, str(pc), str(mesh), "--resolution", "400", "--gpus", str(gpus)], capture_output=True, text=True, cwd=d2_, timeout=300)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
    heads = lambda r: [l.split(":")[0] for l in r.stdout.strip().splitlines()]
    assert heads(r2) == heads(r1)                                         # the reference's twelve lines, same order
    assert np.array_equal(_read_png_rgba(d1 / "texture.png"), _read_png_rgba(d2_ / "texture.png"))
    rows = lambda d: np.array([[float(v) for v in l.split()] for l in open(d / "transfer.ply").read().split("end_header\n")[1].strip().splitlines()[:verts.shape[1]]])
    a, b = rows(d1), rows(d2_)
    assert np.abs(a[:, 8:] - b[:, 8:]).max() <= 1 and np.abs(a[:, :8] - b[:, :8]).max() <= 2e-5
    # the launcher removes its rendezvous directory: none of THIS run's may be left (named in its stderr line, see run_launcher)
    left = [l.split()[-1] for l in r2.stderr.splitlines() if l.startswith("[pt_hip launcher] rendezvous")]
    assert left and not any(os.path.exists(d) for d in left)
    # every rank parsed its own share of the cloud file, and the shares tile it
    parsed = sorted(tuple(int(v) for v in re.findall(r"parsed records \[(\d+), (\d+)\) of (\d+)", l)[0]) for l in r2.stderr.splitlines() if "parsed records" in l)
    assert len(parsed) == gpus and parsed[0][0] == 0 and parsed[-1][1] == parsed[-1][2] == src.shape[1]
    assert all(parsed[i][1] == parsed[i + 1][0] for i in range(gpus - 1))


# ---- refined cells (pt_refine.hip): sub-grids inside heavy cells, descended into by the group kernel ---------------------------
@pytest.mark.parametrize("kind,k,f64,thr", [("blobs", 8, False, 24), ("blobs", 32, True, 64), ("sheet", 20, False, 3), ("lattice", 8, False, 8),
                                            ("line", 16, True, 3), ("uniform", 8, False, 2)])
def test_refined_cells_are_exact(pkg, oracle, kind, k, f64, thr):
    """A low threshold forces nodes everywhere (three levels deep on the clumps): unbounded and radius-bounded searches over the
    refined grid equal brute force bit for bit -- duplicates (lattice: thousands of identical points per node), points on cell
    faces, targets far outside, k above a node's population."""
    import torch
    from test_gpu_stress import _cloud
    rng = np.random.default_rng(77 + k)
    n, m = 60000, 3000
    src = _cloud(rng, kind, n); tgt = _cloud(rng, kind, m)
    tgt[:, :100] = tgt[:, :100] * np.float32(3.0) - np.float32(1.0)            # some targets outside the cloud
    tgt[:, 100:400] = src[:, rng.integers(0, n, 300)]                          # and some exactly on source points
    if f64:
        src = src.astype(np.float64) + (rng.random(src.shape) - 0.5) * 1e-9 * (kind != "lattice"); tgt = tgt.astype(np.float64)
    want = oracle.knn_bruteforce(src, tgt, k)
    for tile in (0, 1):                                                        # group kernel alone / tile kernel + refined leftovers
        with pkg.PointsTransfer(device=0, k_hint=k) as p:
            p.set_param("refine_threshold", thr); p.set_param("tile", tile)
            p.build(src)
            st = p.stats()
            assert st["n_nodes"] > 0 and st["refine_levels"] >= 1 and st["max_cell_points"] > thr, st
            got = p.query(tgt, k)
            _check_exact(got, want, "refined %s tile=%d" % (kind, tile))
            # radius-bounded (what a slab answers for another rank): the bound of every other row halved -> those rows come back shorter
            bnd = want[1][:, k - 1].copy(); bnd[::2] *= 0.5
            xt = pkg.F64 if f64 else pkg.F32
            x = torch.from_numpy(np.ascontiguousarray(tgt)).cuda(); b = torch.from_numpy(bnd).cuda()
            bi = torch.empty((m, k), dtype=torch.int32, device="cuda"); bd = torch.empty((m, k), dtype=torch.float64, device="cuda")
            p.query_bounded_dev(x, xt, b, m, k, bi, bd)
            torch.cuda.synchronize()
            bi = bi.cpu().numpy().view(np.uint32); bd = bd.cpu().numpy()
            keep = want[1] <= bnd[:, None]
            assert np.array_equal(np.where(keep, want[0], 0xFFFFFFFF), bi) and np.array_equal(np.where(keep, want[1], np.inf), bd)
    with pkg.PointsTransfer(device=0, k_hint=k) as p:                           # the refinement changes nothing but the speed
        p.set_param("refine_threshold", 0)
        p.build(src)
        assert p.stats()["n_nodes"] == 0
        _check_exact(p.query(tgt, k), want, "unrefined %s" % kind)


@pytest.mark.parametrize("kind,k,f64,thr,rho,wmin", [("blobs", 8, False, 0, None, 1), ("blobs", 32, True, 48, None, 1), ("sheet", 20, False, 3, None, 2), ("lattice", 8, False, 0, None, 2),
                                                     ("lattice", 32, False, 32, None, 1), ("line", 16, True, 3, None, 1), ("uniform", 32, False, 0, 0.4, 1), ("uniform", 8, True, 0, None, 2),
                                                     ("blobs", 20, False, 24, None, 2)])
def test_wave_kernel_is_exact(pkg, oracle, kind, k, f64, thr, rho, wmin):
    """One wave per target (knn_wave_kernel), forced for every target (or every target with two points in its 27 nearest cells): dense cells,
    refined cells (thr > 0: descended into, 64 rows per step), sparse neighbourhoods that need several shells (rho 0.4 at k = 32), a
    target next to a lone outlier (block sweep), radius-bounded queries -- all equal to brute force bit for bit."""
    import torch
    from test_gpu_stress import _cloud
    rng = np.random.default_rng(900 + k)
    n, m = 50000, 2500
    src = _cloud(rng, kind, n); tgt = _cloud(rng, kind, m)
    if kind in ("blobs", "lattice"):
        src[:, -1] = np.float32(40.0)                                          # a lone point far away ...
        tgt[:, 0] = np.float32(40.0) + np.float32(1e-3)                        # ... and a target beside it: everything else is 40 units off
    tgt[:, 1:60] = tgt[:, 1:60] * np.float32(2.0) - np.float32(0.5)            # some targets around the cloud's edge
    tgt[:, 100:300] = src[:, rng.integers(0, n, 200)]
    if f64:
        src = src.astype(np.float64) + (rng.random(src.shape) - 0.5) * 1e-9 * (kind != "lattice"); tgt = tgt.astype(np.float64)
    want = oracle.knn_bruteforce(src, tgt, k)
    with pkg.PointsTransfer(device=0, k_hint=min(k, 32), rho=rho) as p:
        # wave_min 1: no group kernel at all (targets next to refined cells found by the per-cell flag); 2: the group kernel marks the targets
        p.set_param("refine_threshold", thr); p.set_param("tile", 0); p.set_param("wave_force", 1); p.set_param("wave_min", wmin)
        p.build(src)
        st = p.stats()
        assert (st["n_nodes"] > 0) == (thr > 0), st
        got = p.query(tgt, k)
        assert p.stats()["n_wave"] >= m // 4, p.stats()                          # (targets with 27 empty cells around them stay with the group kernel)
        _check_exact(got, want, "wave %s" % kind)
        bnd = want[1][:, k - 1].copy(); bnd[::2] *= 0.5
        xt = pkg.F64 if f64 else pkg.F32
        x = torch.from_numpy(np.ascontiguousarray(tgt)).cuda(); b = torch.from_numpy(bnd).cuda()
        bi = torch.empty((m, k), dtype=torch.int32, device="cuda"); bd = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_bounded_dev(x, xt, b, m, k, bi, bd)
        torch.cuda.synchronize()
        bi = bi.cpu().numpy().view(np.uint32); bd = bd.cpu().numpy()
        keep = want[1] <= bnd[:, None]
        assert np.array_equal(np.where(keep, want[0], 0xFFFFFFFF), bi) and np.array_equal(np.where(keep, want[1], np.inf), bd)


@pytest.mark.parametrize("f64", [False, True])
def test_runs_of_identical_points_keep_their_lowest_indices_in_front(pkg, oracle, f64):
    """Quantised clouds pile thousands of points on one position.  Leaves of refined cells that hold ONE position more than 32 times are
    rewritten -- 32 lowest indices first, ascending -- and tagged (pt_stats.dup_leaves); the wave kernel reads the front only, the group
    kernel the whole leaf, a flat scan the whole cell: all three equal brute force under (d2, index), at k = 32 (every front entry may be
    needed), k = 5, and under radius bounds; "dup_runs" 0 gives the same answers without the tags."""
    import torch
    rng = np.random.default_rng(411)
    nodes = rng.random((3, 40)).astype(np.float32)                                      # 40 positions ...
    mult = rng.integers(1, 900, 40)                                                     # ... 1 to 900 points each
    src = np.repeat(nodes, mult, axis=1)
    src = np.concatenate([src, rng.random((3, 20000)).astype(np.float32)], axis=1)      # and a thin background
    src = src[:, rng.permutation(src.shape[1])]                                         # duplicates carry scattered indices
    n = src.shape[1]
    tgt = np.concatenate([nodes[:, :30] + np.float32(1e-4) * rng.standard_normal((3, 30)).astype(np.float32), nodes[:, 10:40], rng.random((3, 500)).astype(np.float32)], axis=1)
    if f64:
        src = src.astype(np.float64); tgt = tgt.astype(np.float64)
    m = tgt.shape[1]
    for k in (32, 5):
        want = oracle.knn_bruteforce(src, tgt, k)
        for dup, wmin, tile in ((1, 1, 0), (1, 0, 0), (0, 1, 0), (1, 1, 1)):
            with pkg.PointsTransfer(device=0, k_hint=k) as p:
                p.set_param("refine_threshold", 64); p.set_param("dup_runs", dup); p.set_param("tile", tile); p.set_param("wave_force", 1); p.set_param("wave_min", wmin)
                p.build(src)
                st = p.stats()
                assert st["n_nodes"] > 0 and (st["dup_leaves"] > 0) == (dup == 1), st
                _check_exact(p.query(tgt, k), want, "dup=%d wave_min=%d tile=%d k=%d" % (dup, wmin, tile, k))
                if dup and wmin and not tile:
                    bnd = want[1][:, k - 1].copy(); bnd[::2] *= 0.5
                    x = torch.from_numpy(np.ascontiguousarray(tgt)).cuda(); b = torch.from_numpy(bnd).cuda()
                    bi = torch.empty((m, k), dtype=torch.int32, device="cuda"); bd = torch.empty((m, k), dtype=torch.float64, device="cuda")
                    p.query_bounded_dev(x, pkg.F64 if f64 else pkg.F32, b, m, k, bi, bd)
                    torch.cuda.synchronize()
                    keep = want[1] <= bnd[:, None]
                    assert np.array_equal(np.where(keep, want[0], 0xFFFFFFFF), bi.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("k,f64", [(20, False), (8, False), (24, True), (32, False)])
def test_tile_kernel_over_the_blocks_that_hold_targets(pkg, oracle, k, f64):
    """A surface leaves most of its grid empty: the tile kernel is then launched over a LIST of the blocks that hold targets ("tile_sparse"),
    in the two-workgroups-per-CU geometries first -- 384 threads for k in 17..24 -- with the over-budget blocks retried in the large one,
    and ("tile_contrast" 1) before the wave kernel on clouds whose occupied cells stay far above rho.  Same neighbours as brute force."""
    rng = np.random.default_rng(77 + k)
    n, m = 400_000, 6000
    v = rng.standard_normal((3, n)); v /= np.linalg.norm(v, axis=0, keepdims=True)
    src = (0.5 + 0.45 * v + 1e-4 * rng.standard_normal((3, n)))
    w = rng.standard_normal((3, m)); w /= np.linalg.norm(w, axis=0, keepdims=True)
    tgt = (0.5 + 0.45 * w + 1e-3 * rng.standard_normal((3, m)))
    src = src.astype(np.float64 if f64 else np.float32); tgt = tgt.astype(src.dtype)
    want = oracle.knn_bruteforce(src, tgt, k)
    for sparse, contrast in ((1, 1), (0, 1), (2, 0)):
        with pkg.PointsTransfer(device=0, k_hint=k) as p:
            p.set_param("tile_sparse", sparse); p.set_param("tile_contrast", contrast)
            p.build(src)
            got = p.query(tgt, k)
            st = p.stats()
            _check_exact(got, want, "sparse=%d contrast=%d" % (sparse, contrast))
            if contrast and k <= 24:
                assert st["n_leftover"] < m, st                                  # the tile kernel settled some of them


def test_config5_shape_sampled_against_oracle(pkg, oracle):
    """BASELINE config 5's own parameters at a tenth of its size: 30 M clustered points with fp16 coordinates (a lattice: hundreds of
    exact duplicates per node in the clumps), 1.5 M targets, k = 32 -- tile kernel, wave kernel and its descending variant; a
    sample is exact against the CPU kd-tree (ties among duplicates resolved by index)."""
    import torch
    n, m, k, seed = 30_000_000, 1_500_000, 32, 0xC5
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build_synth(n, seed, dist=pkg.capi.DIST_CLUSTERED, xyz_type=pkg.F16)
        p.targets_synth(m, seed, dist=pkg.capi.DIST_CLUSTERED, xyz_type=pkg.F16)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        st = p.stats()
        assert st["n_wave"] > 0 and st["n_nodes"] > 0, st        # (at this size the tile kernel takes part of the targets, the wave kernel the rest)
        assert bool((d2[:, 1:] >= d2[:, :-1]).all())
        sel = np.random.default_rng(4).choice(m, 3000, replace=False)
        I = idx[torch.from_numpy(sel).cuda()].cpu().numpy().view(np.uint32); D = d2[torch.from_numpy(sel).cuda()].cpu().numpy()
    src = oracle.synth_xyz(seed, 0, n, dist=1, n_total=n, m_total=m).astype(np.float16).astype(np.float32)
    tgt = np.concatenate([oracle.synth_xyz(seed, 1, 1, i0=int(t), dist=1, n_total=n, m_total=m) for t in sel], axis=1).astype(np.float16).astype(np.float32)
    wi, wd = oracle.KdTree(src).query(tgt, k)
    assert np.array_equal(I, wi) and np.array_equal(D, wd)


@pytest.mark.parametrize("kind,f64", [("uniform", False), ("blobs", False), ("sheet", True)])
def test_three_level_sort_for_very_fine_grids(pkg, oracle, kind, f64):
    """More than 1024 macro blocks (64^3 cells each): the sort partitions by GROUPS of macro blocks first, then by macro block, then
    by block -- one pass more than usual.  Forced here with a tiny rho (a grid of ~1e9 cells over 300 k points); searches through the
    tile kernel and through the group kernel equal brute force bit for bit."""
    from test_gpu_stress import _cloud
    rng = np.random.default_rng(31)
    n, m, k = 300000, 2000, 8
    src = _cloud(rng, kind, n); tgt = _cloud(rng, kind, m)
    tgt[:, :50] = tgt[:, :50] * np.float32(1.5) - np.float32(0.25)
    if f64:
        src = src.astype(np.float64) + (rng.random(src.shape) - 0.5) * 1e-9; tgt = tgt.astype(np.float64)
    want = oracle.knn_bruteforce(src, tgt, k)
    for tile in (0, 1):
        with pkg.PointsTransfer(device=0, rho=0.0004) as p:
            p.set_param("adaptive", 0); p.set_param("tile", tile)
            p.build(src)
            st = p.stats()
            assert st["n_levels"] == 3, st
            _check_exact(p.query(tgt, k), want, "three-level %s tile=%d" % (kind, tile))


def test_rebuild_starts_from_the_remembered_cell_size(pkg, oracle):
    """pt_rebuild of the same resident cloud: the cell size the first build searched for (refinement steps, one full sort each) is where
    the rebuild starts, checked against the occupancy it finds; results identical; a new cloud in the same context searches again."""
    from test_gpu_stress import _cloud
    rng = np.random.default_rng(5)
    src = _cloud(rng, "blobs", 200000); tgt = _cloud(rng, "blobs", 2000)
    want = oracle.knn_bruteforce(src, tgt, 8)
    with pkg.PointsTransfer(device=0, k_hint=8) as p:
        p.build(src)
        st0 = p.stats()
        assert st0["n_refine"] >= 1
        _check_exact(p.query(tgt, 8), want, "first build")
        p.rebuild()
        st1 = p.stats()
        assert st1["grid_dim"] == st0["grid_dim"] and st1["n_refine"] == st0["n_refine"]
        _check_exact(p.query(tgt, 8), want, "rebuild from the remembered cell size")
        uni = rng.random((3, 150000), dtype=np.float32)
        p.build(uni)                                          # another cloud: no hint
        assert p.stats()["n_refine"] == 0
        _check_exact(p.query(tgt, 8), oracle.knn_bruteforce(uni, tgt, 8), "new cloud")


def test_full_size_clustered_sampled_against_oracle(pkg, oracle):
    """BASELINE config 5's distribution at 100 M points (thin patches + blobs + a little uniform; cells of the fullest clump hold tens
    of thousands of points and get sub-grids): the search of 5 M jittered targets is exact on a sample against the CPU kd-tree."""
    import torch
    n, m, k, seed = 100_000_000, 5_000_000, 8, 0xC5
    with pkg.PointsTransfer(device=0, k_hint=k) as p:
        p.build_synth(n, seed, dist=pkg.capi.DIST_CLUSTERED)
        p.targets_synth(m, seed, dist=pkg.capi.DIST_CLUSTERED)
        st = p.stats()
        assert st["n_nodes"] > 0 and st["max_cell_points"] > 2048
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        p.query_resident_dev(k, idx, d2)
        torch.cuda.synchronize()
        assert bool((d2[:, 1:] >= d2[:, :-1]).all())
        sel = np.random.default_rng(2).choice(m, 4000, replace=False)
        I = idx[torch.from_numpy(sel).cuda()].cpu().numpy().view(np.uint32); D = d2[torch.from_numpy(sel).cuda()].cpu().numpy()
    src = oracle.synth_xyz(seed, 0, n, dist=1, n_total=n, m_total=m)
    tgt = np.concatenate([oracle.synth_xyz(seed, 1, 1, i0=int(t), dist=1, n_total=n, m_total=m) for t in sel], axis=1)
    wi, wd = oracle.KdTree(src).query(tgt, k)
    assert np.array_equal(I, wi) and np.array_equal(D, wd)
