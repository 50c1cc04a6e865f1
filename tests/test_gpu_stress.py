"""Randomised differential test of the HIP k-NN path against the CPU oracle: many small configurations drawn from the
corners the kernels branch on (densities that overflow the LDS tile, rows longer than one DMA, exact ties on a lattice,
sheets and lines, targets far outside the cloud, k above the point count, every tile geometry).  PT_STRESS_CASES sets the
number of cases (default 24; a longer run is a one-off, e.g. PT_STRESS_CASES=400)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cloud(rng, kind, n):
    if kind == "uniform":
        return rng.random((3, n), dtype=np.float32)
    if kind == "lattice":                      # exact distance ties everywhere
        g = int(rng.integers(3, 40))
        return (rng.integers(0, g, size=(3, n)).astype(np.float32) / np.float32(g))
    if kind == "blobs":                        # dense clumps in a sparse background: overflowing tiles, long rows and cells
        c = rng.random((3, int(rng.integers(1, 6))), dtype=np.float32)
        s = np.float32(10.0 ** rng.uniform(-4, -1.5))
        p = c[:, rng.integers(0, c.shape[1], size=n)] + s * rng.standard_normal((3, n)).astype(np.float32)
        bg = rng.random((3, n), dtype=np.float32)
        keep = rng.random(n) < 0.85
        return np.where(keep, p, bg).astype(np.float32)
    if kind == "sheet":
        p = rng.random((3, n), dtype=np.float32)
        p[int(rng.integers(0, 3))] = np.float32(0.37) + np.float32(1e-5) * rng.standard_normal(n).astype(np.float32)
        return p
    if kind == "line":
        t = rng.random(n, dtype=np.float32)
        return np.stack([t, np.float32(0.5) + np.float32(1e-6) * t, np.float32(0.25) * t]).astype(np.float32)
    raise AssertionError(kind)


@pytest.mark.parametrize("case", range(int(os.environ.get("PT_STRESS_CASES", "24"))))
def test_random_configuration_matches_oracle(pkg, oracle, case):
    rng = np.random.default_rng(1000 + case)
    kind = ["uniform", "lattice", "blobs", "sheet", "line"][case % 5]
    n = int(rng.choice([1, 7, 300, 5000, 60000, 250000]))
    m = int(rng.choice([0, 1, 33, 2000, 15000]))
    k = int(rng.choice([1, 2, 5, 8, 9, 16, 17, 20, 24, 25, 32]))
    rho = rng.choice([0.0, 1.0, 2.5, 6.0, 12.0, 40.0])            # 0: the k_hint policy
    tile = int(rng.choice([1, 1, 2, 3, 0]))
    src = _cloud(rng, kind, n)
    tgt = _cloud(rng, kind, m) if m else np.zeros((3, 0), np.float32)
    if m:
        far = rng.random(m) < 0.05                                        # some targets well outside the cloud
        tgt[:, far] = tgt[:, far] * np.float32(4.0) - np.float32(1.5)
        near = rng.random(m) < 0.3                                        # and some right on top of source points
        if n:
            tgt[:, near] = src[:, rng.integers(0, n, size=int(near.sum()))]
    if rng.random() < 0.4:            # clouds far from the origin / tiny / huge: coarse fp32 spacing, many exact ties
        sc = np.float32(10.0 ** rng.uniform(-4, 4)); off = np.float32(rng.choice([0.0, 1.0, -37.5, 1000.0, 65536.0]) * float(sc))
        src = (src * sc + off).astype(np.float32); tgt = (tgt * sc + off).astype(np.float32)
    kw = dict(k_hint=k) if rho == 0.0 else dict(rho=float(rho))
    f64 = rng.random() < float(os.environ.get("PT_STRESS_F64", "0.3"))    # the double-precision path (fp32 shadow in LDS, exact 32-byte records)
    if f64:
        src = src.astype(np.float64) + (rng.random(src.shape) - 0.5) * 1e-9 * (kind != "lattice")
        tgt = tgt.astype(np.float64)
    # routing knobs (drawn last, so that the clouds of earlier rounds' cases stay what they were): who goes to the one-wave-per-target
    # kernel, how heavy a cell must be to get a sub-grid, how fine the refinement of the cell size may go
    wforce = int(rng.random() < 0.4); wmin = int(rng.choice([1, 1, 2, 300])); thr = int(rng.choice([8192, 8192, 0, 4, 40]))
    cpp = float(rng.choice([2.0, 2.0, 0.5, 16.0]))
    # round 4's knobs, again drawn after everything older: the tile kernel ahead of the wave kernel on clouds with density contrast, the tile
    # kernel over the list of target-holding blocks, the fronts of leaves that hold one position many times (lattice clouds at thr 4 / 40)
    tcon = int(rng.random() < 0.4); tsparse = int(rng.choice([2, 2, 1, 0])); dupr = int(rng.random() < 0.8)
    with pkg.PointsTransfer(device=0, **kw) as p:
        p.set_param("tile", tile); p.set_param("wave_force", wforce); p.set_param("wave_min", wmin); p.set_param("refine_threshold", thr)
        p.set_param("refine_cells_per_point", cpp)
        p.set_param("tile_contrast", tcon); p.set_param("tile_sparse", tsparse); p.set_param("dup_runs", dupr)
        p.build(src, xyz_type=pkg.F64 if f64 else None)
        gi, gd = p.query(tgt, k, xyz_type=pkg.F64 if f64 else None)
        if case % 3 == 0:                                                  # a rebuild starts from the remembered cell size: same answer
            p.rebuild()
            gi2, gd2 = p.query(tgt, k, xyz_type=pkg.F64 if f64 else None)
            assert np.array_equal(gi, gi2) and np.array_equal(gd, gd2), "case %d: rebuild changed the answer" % case
    wi, wd = oracle.KdTree(src).query(tgt, k) if n else (np.full((m, k), 0xFFFFFFFF, np.uint32), np.full((m, k), np.inf))
    what = "case %d: %s n=%d m=%d k=%d rho=%g tile=%d f64=%d wave=%d/%d thr=%d cpp=%g tcon=%d tsparse=%d dup=%d" % (case, kind, n, m, k, rho, tile, f64, wforce, wmin, thr, cpp, tcon, tsparse, dupr)
    assert np.array_equal(gi, wi), what + ": indices differ in %d rows" % int((gi != wi).any(axis=1).sum())
    assert np.array_equal(gd, wd), what + ": d2 differ"


@pytest.mark.parametrize("case", range(int(os.environ.get("PT_STRESS_BLEND_CASES", "10"))))
def test_random_fused_blend_matches_oracle(pkg, oracle, case):
    """The fused k-NN + blend call on generated clouds (uniform and clustered, both blend modes, every tile geometry):
    neighbours bit-exact, blended colour / normal within the path's 1e-5."""
    import torch
    rng = np.random.default_rng(5000 + case)
    n = int(rng.choice([2000, 40000, 300000])); m = int(rng.choice([1, 500, 12000]))
    k = int(rng.choice([1, 4, 8, 13, 16, 20, 27, 32])); mode = int(rng.integers(0, 2)); tile = int(rng.choice([1, 2, 3, 0]))
    dist = int(rng.integers(0, 2)); seed = int(rng.integers(1, 1 << 30))
    rho = rng.choice([0.0, 2.0, 5.0, 15.0])
    kw = dict(k_hint=k) if rho == 0.0 else dict(rho=float(rho))
    with pkg.PointsTransfer(device=0, **kw) as p:
        p.set_param("tile", tile)
        p.build_synth(n, seed, dist=dist)
        p.targets_synth(m, seed, dist=dist)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); d2 = torch.empty((m, k), dtype=torch.float64, device="cuda")
        rgb = torch.empty((m, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty((m, 3), dtype=torch.float32, device="cuda")
        p.query_blend_resident_dev(k, mode, idx, d2, rgb, nrm)
        torch.cuda.synchronize()
    src = oracle.synth_xyz(seed, 0, n, dist=dist, n_total=n, m_total=m); tgt = oracle.synth_xyz(seed, 1, m, dist=dist, n_total=n, m_total=m)
    wi, wd = oracle.KdTree(src).query(tgt, k)
    what = "case %d: n=%d m=%d k=%d mode=%d tile=%d dist=%d rho=%g" % (case, n, m, k, mode, tile, dist, rho)
    assert np.array_equal(idx.cpu().numpy().view(np.uint32), wi) and np.array_equal(d2.cpu().numpy(), wd), what
    rc, rn = oracle.blend(wi, wd, oracle.synth_rgb(seed, n), oracle.synth_nrm(seed, n), mode)
    assert np.abs(rgb.cpu().numpy() - rc).max() / 255 <= 1e-5 and np.abs(nrm.cpu().numpy() - rn).max() <= 1e-5, what


@pytest.mark.parametrize("case", range(int(os.environ.get("PT_STRESS_POOL_CASES", "8"))))
def test_random_two_level_pooled_builds_match_oracle(pkg, oracle, case):
    """The two-level sort with BOTH histogram passes replaced by estimates (round 3: bin regions from a sample for pass 1, block regions
    from the macro counts for pass 2 on rebuilds) on clouds small enough for the oracle: whatever the cloud -- uniform, lattice, blobs,
    a sheet -- every build either keeps inside its regions or notices (pt_stats.pass1_pooled / pass2_pooled = -1) and redoes the pass
    exactly; the answer is the oracle's after the first build and after two rebuilds."""
    rng = np.random.default_rng(9000 + case)
    kind = ["uniform", "blobs", "lattice", "sheet"][case % 4]
    n = int(rng.choice([700_000, 1_500_000])); m = int(rng.choice([500, 6000])); k = int(rng.choice([1, 8, 16, 20]))
    rho = float(rng.choice([0.5, 1.0]))                                  # fine cells: more than 1024 blocks, i.e. the two-level sort
    f64 = rng.random() < 0.3
    src = _cloud(rng, kind, n); tgt = _cloud(rng, kind, m)
    near = rng.random(m) < 0.3                                           # some targets right on top of source points
    tgt[:, near] = src[:, rng.integers(0, n, size=int(near.sum()))]
    if f64:
        src = src.astype(np.float64) + (rng.random(src.shape) - 0.5) * 1e-9 * (kind != "lattice"); tgt = tgt.astype(np.float64)
    seen = []
    with pkg.PointsTransfer(device=0, rho=rho) as p:
        p.set_param("pool_min_points", 1)
        if kind == "uniform":
            p.set_param("refine_cells_per_point", 16.0)
        p.build(src, xyz_type=pkg.F64 if f64 else None)
        st = p.stats(); seen.append((st["n_levels"], st["pass1_pooled"], st["pass2_pooled"]))
        res = [p.query(tgt, k, xyz_type=pkg.F64 if f64 else None)]
        for _ in range(2):
            p.rebuild()
            st = p.stats(); seen.append((st["n_levels"], st["pass1_pooled"], st["pass2_pooled"]))
            res.append(p.query(tgt, k, xyz_type=pkg.F64 if f64 else None))
    wi, wd = oracle.KdTree(src).query(tgt, k)
    what = "case %d: %s n=%d m=%d k=%d rho=%g f64=%d (levels, pass 1, pass 2 per build: %s)" % (case, kind, n, m, k, rho, f64, seen)
    for gi, gd in res:
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd), what
    if kind == "uniform":
        assert seen[0][0] == 2 and seen[0][1] == 1, what        # the pooled pass 1 did run (pass 2 is pooled only where the build calls the cloud uniform)
