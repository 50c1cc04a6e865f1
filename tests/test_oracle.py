"""CPU suite, part 1: the oracle against its golden vectors, against itself (brute force vs kd-tree
restatement vs numpy vs scipy) and against the reference-side facts that ARE pinned (Point layout from
the reference's own header, Distance.h known answers recorded in SURVEY.md 8c)."""
import numpy as np
import pytest


def test_generator_kats(oracle, golden):
    assert np.array_equal(oracle.synth_xyz(0xC1, 0, 8), golden["kat_src_xyz"])
    assert np.array_equal(oracle.synth_xyz(0xC1, 1, 8), golden["kat_tgt_xyz"])
    assert np.array_equal(oracle.synth_rgb(0xC1, 8), golden["kat_rgb"])
    assert np.array_equal(oracle.synth_nrm(0xC1, 8), golden["kat_nrm"])
    # SplitMix64 published test vector: seed 0 -> first output 0xE220A8397B1DCDAF
    assert oracle.lib().pto_splitmix64(0) == 0xE220A8397B1DCDAF
    x = oracle.synth_xyz(0xC2, 0, 1000)
    assert x.min() >= 0 and x.max() < 1 and np.array_equal(x, np.round(x * 2**24) / 2**24)   # 24-bit lattice
    # index-addressable: a window equals the same window of a longer run
    assert np.array_equal(oracle.synth_xyz(0xC2, 0, 100, i0=500), x[:, 500:600])
    n = oracle.synth_nrm(0xC2, 1000)
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)


def test_clustered_generator_kats(oracle, golden):
    assert np.array_equal(oracle.synth_xyz(0xC5, 0, 8, dist=1), golden["kat_clustered_src"])
    assert np.array_equal(oracle.synth_xyz(0xC5, 1, 8, dist=1, n_total=1000, m_total=100), golden["kat_clustered_tgt"])


def test_clustered_generator_shape(oracle):
    n, m = 50000, 2500
    src = oracle.synth_xyz(0xC5, 0, n, dist=1)
    tgt = oracle.synth_xyz(0xC5, 1, m, dist=1, n_total=n, m_total=m)
    assert src.min() >= 0 and src.max() < 1 and tgt.min() >= 0 and tgt.max() < 1
    # clustered: the occupancy of a 32^3 grid is far below what uniform points would give
    occ = len(np.unique(np.array([1, 32, 1024]) @ (src * 32).astype(np.int64)))
    uni = len(np.unique(np.array([1, 32, 1024]) @ (oracle.synth_xyz(0xC5, 0, n) * 32).astype(np.int64)))
    assert occ < 0.6 * uni
    # targets sit next to the sources they were drawn from (jitter 5e-4 * unit-variance noise)
    step = n // m
    d = np.abs(tgt - src[:, (np.arange(m) * step) % n]).max()
    assert d < 5e-3
    # index-addressable
    assert np.array_equal(oracle.synth_xyz(0xC5, 0, 100, i0=777, dist=1), src[:, 777:877])


def test_distance_known_answers(oracle):
    # reference src/Distance.h:6-11, :27-57, :60-90, :92-95, :97, :99 -- values from SURVEY.md 8c
    assert oracle.transformed_distance([0, 0, 0], [1, 2, 3]) == 14.0
    d, off = oracle.min_distance_to_rectangle([.5, 3, .5], [0, 0, 0], [1, 1, 1])
    assert d == 4.0 and list(off) == [0, 2, 0]
    d, off = oracle.max_distance_to_rectangle([.5, 3, .5], [0, 0, 0], [1, 1, 1])
    assert d == 9.5 and list(off) == [.5, 3, .5]
    assert oracle.new_distance(10.0, 1.0, 3.0) == 18.0
    L = oracle.lib()
    assert L.pto_transformed_distance_scalar(3.0) == 9.0 and L.pto_inverse_of_transformed_distance(9.0) == 3.0


def test_metric_is_unfused(oracle):
    # (dx*dx + dy*dy) + dz*dz with every op rounded: compare with numpy's op-by-op float64
    rng = np.random.default_rng(7)
    p = rng.random((1000, 3)); q = rng.random((1000, 3))
    for a, b in zip(p, q):
        dx, dy, dz = a - b
        assert oracle.transformed_distance(a, b) == (dx * dx + dy * dy) + dz * dz


def test_reference_point_layout(oracle):
    # produced by oracle/_ref/point_layout, compiled from the reference's own src/Point.h
    lay = oracle.ref_point_layout()
    assert lay["sizeof"] == 80 and lay["alignof"] == 8
    assert (lay["off_ver"], lay["off_normal"], lay["off_color"], lay["off_U"], lay["off_V"]) == (0, 24, 48, 64, 72)
    assert lay["trivially_copyable"] == 1 and lay["standard_layout"] == 1
    assert lay["eq_xyz_only"] == 1 and lay["default_ver_zero"] == 1 and lay["coord_begin_is_ver"] == 1 and lay["coord_len"] == 3


@pytest.mark.parametrize("name,ks", [("c1", (1, 8, 16, 32)), ("ties", (1, 8, 20)), ("outside", (8,)), ("tiny", (8, 32)), ("flat", (8,))])
def test_oracle_matches_golden_and_kdtree(oracle, golden, golden_cases, name, ks):
    src, tgt = golden_cases[name]
    kd = oracle.KdTree(src)
    for k in ks:
        sub = tgt[:, :256] if (name == "c1" and k > 8) else tgt
        idx, d2 = oracle.knn_bruteforce(src, sub, k)
        assert np.array_equal(idx, golden["%s_k%d_idx" % (name, k)])
        assert np.array_equal(d2, golden["%s_k%d_d2" % (name, k)])
        ik, dk = kd.query(sub, k)                       # the kd-tree restatement = the timed CPU baseline
        assert np.array_equal(ik, idx) and np.array_equal(dk, d2)
    kd.close()


def test_oracle_third_opinions(oracle, golden_cases):
    src, tgt = golden_cases["c1"]
    idx, d2 = oracle.knn_bruteforce(src, tgt[:, :64], 8)
    inn, dnn = oracle.knn_numpy(src, tgt[:, :64], 8)
    assert np.array_equal(idx, inn) and np.array_equal(d2, dnn)
    from scipy.spatial import cKDTree
    dd, ii = cKDTree(src.T.astype(np.float64)).query(tgt.T.astype(np.float64), k=8)
    idx, d2 = oracle.knn_bruteforce(src, tgt, 8)
    same = np.sort(ii, axis=1) == np.sort(idx.astype(np.int64), axis=1)     # neighbour SETS (tie-free rows)
    assert same.all(axis=1).mean() > 0.99


def test_oracle_ties_and_missing(oracle, golden_cases):
    src, tgt = golden_cases["ties"]
    idx, d2 = oracle.knn_bruteforce(src, tgt, 20)
    assert (np.diff(d2, axis=1) >= 0).all()
    tie = np.diff(d2, axis=1) == 0
    assert tie.any() and (np.diff(idx.astype(np.int64), axis=1)[tie] > 0).all()   # equal d2 -> ascending index
    src, tgt = golden_cases["tiny"]
    idx, d2 = oracle.knn_bruteforce(src, tgt, 8)
    assert (idx[:, 5:] == oracle.NOIDX).all() and np.isinf(d2[:, 5:]).all() and (idx[:, :5] < 5).all()


def test_oracle_global_index_and_merge(oracle):
    rng = np.random.default_rng(3)
    src = rng.random((3, 3000)).astype(np.float32); tgt = rng.random((3, 200)).astype(np.float32)
    k = 8
    full_i, full_d = oracle.knn_bruteforce(src, tgt, k)
    order = np.argsort(src[0], kind="stable")
    parts = np.array_split(order, 3)                         # three x-slabs with global indices
    li = np.stack([oracle.knn_bruteforce(src[:, p], tgt, k, gidx=p.astype(np.uint32))[0] for p in parts])
    ld = np.stack([oracle.knn_bruteforce(src[:, p], tgt, k, gidx=p.astype(np.uint32))[1] for p in parts])
    mi, md = oracle.merge_candidates(li, ld)
    assert np.array_equal(mi, full_i) and np.array_equal(md, full_d)


def test_oracle_blend_and_pca(oracle, golden, golden_cases):
    src, tgt = golden_cases["c1"]
    rgb, nrm = oracle.synth_rgb(0xC1, 10000), oracle.synth_nrm(0xC1, 10000)
    idx, d2 = golden["c1_k8_idx"], golden["c1_k8_d2"]
    for mode in (0, 1):
        c, n = oracle.blend(idx, d2, rgb, nrm, mode)
        assert np.array_equal(c, golden["c1_k8_blend%d_rgb" % mode]) and np.array_equal(n, golden["c1_k8_blend%d_nrm" % mode])
    c, n = oracle.blend(idx, d2, rgb, nrm, 0)
    assert np.allclose(c, rgb[idx].astype(np.float64).mean(axis=1), atol=1e-4)
    pn, plan = oracle.pca_normals(golden["c1_k16_idx"], src, nrm)
    assert np.allclose(pn, golden["c1_k16_pca_nrm"], atol=1e-6) and np.allclose(plan, golden["c1_k16_pca_planarity"], atol=1e-9)
    # PCA: points on the plane z = 0.3 + 0.1x must give normal ~ (-0.1, 0, 1)/|.|
    rng = np.random.default_rng(5)
    p = rng.random((3, 2000)); p[2] = 0.3 + 0.1 * p[0]
    t = p[:, :50]
    i16, _ = oracle.knn_bruteforce(p, t, 16)
    nn, plan = oracle.pca_normals(i16, p)
    want = np.array([-0.1, 0, 1.0]) / np.linalg.norm([-0.1, 0, 1.0])
    assert np.allclose(nn, want, atol=1e-6) and (plan < 1e-12).all()


def test_kdtree_exact_ties_at_the_pruning_bound(oracle):
    """Lattice clouds make the kd-tree's incremental bound EQUAL to the current k-th distance; rounded an ulp high it
    used to prune a subtree holding a tied point with a lower index (caught by tests/test_gpu_stress.py, case 171: the
    GPU agreed with the brute force, the kd-tree did not).  The kd-tree restatement must agree with the brute force on
    such data, far-away targets included."""
    from test_gpu_stress import _cloud
    rng = np.random.default_rng(1000 + 171)                      # the generator sequence of that case
    n = int(rng.choice([1, 7, 300, 5000, 60000, 250000])); m = int(rng.choice([0, 1, 33, 2000, 15000]))
    k = int(rng.choice([1, 2, 5, 8, 9, 16, 17, 20, 24, 25, 32])); rng.choice([0.0, 1.0, 2.5, 6.0, 12.0, 40.0]); rng.choice([1, 1, 2, 3, 0])
    assert (n, m, k) == (250000, 2000, 17)
    src = _cloud(rng, "lattice", n); tgt = _cloud(rng, "lattice", m)
    far = rng.random(m) < 0.05
    tgt[:, far] = tgt[:, far] * np.float32(4.0) - np.float32(1.5)
    sub = tgt[:, far]
    bi, bd = oracle.knn_bruteforce(src, sub, k)
    ki, kd = oracle.KdTree(src).query(sub, k)
    assert np.array_equal(ki, bi) and np.array_equal(kd, bd)
    # and a few more lattices with far targets
    rng = np.random.default_rng(1171)
    for g, n, k in ((28, 60000, 17), (7, 20000, 32), (13, 40000, 8)):
        src = (rng.integers(0, g, size=(3, n)).astype(np.float32) / np.float32(g))
        tgt = np.concatenate([np.array([[-0.5, 2.5, 0.5], [-0.5, -0.5, 1.5], [1.5, 0.5, -2.0]], np.float32),
                              rng.random((3, 100), dtype=np.float32) * 4 - 1.5], axis=1)
        bi, bd = oracle.knn_bruteforce(src, tgt, k)
        ki, kd = oracle.KdTree(src).query(tgt, k)
        assert np.array_equal(ki, bi) and np.array_equal(kd, bd), "lattice %d" % g


def test_reference_mix_formula_known_answer(oracle):
    """pointsTransfer.cpp:95-97 restated (oracle.blend_weighted): double weight x int colour, left-to-right double sum,
    one rounding to float -- checked against the same expression written out in numpy float64."""
    rgb = np.array([[255, 0, 10], [0, 255, 20], [7, 9, 250]], np.uint8)
    nrm = np.array([[0, 0, 1], [0, 1, 0], [1, 0, 0]], np.float32)
    bc = np.array([[0.2, 0.3, 0.5], [1.0 / 3, 1.0 / 3, 1.0 / 3], [1.0, 0.0, 0.0]])
    idx = np.tile(np.arange(3, dtype=np.uint32), (3, 1))
    got, gn = oracle.blend_weighted(idx, bc, rgb, nrm)
    for t in range(3):
        for c in range(3):
            want = np.float32((bc[t, 0] * float(rgb[0, c]) + bc[t, 1] * float(rgb[1, c])) + bc[t, 2] * float(rgb[2, c]))
            assert got[t, c] == want
    assert np.array_equal(got[2], rgb[0].astype(np.float32)) and got[0, 0] == np.float32(0.2 * 255 + 0.0 + 0.5 * 7)
    assert int(got[0, 2]) == 133                              # 0.2*10 + 0.3*20 + 0.5*250; :100-102 truncate the float into an unsigned char
    # a missing neighbour contributes nothing
    idx2 = idx.copy(); idx2[0, 1] = 0xFFFFFFFF
    g2, _ = oracle.blend_weighted(idx2, bc, rgb, nrm)
    assert g2[0, 1] == np.float32(0.2 * 0 + 0.5 * 9)
