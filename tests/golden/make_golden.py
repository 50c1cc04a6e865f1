"""Generates the golden fixtures under tests/golden/ from the CPU oracle (oracle/pt_oracle.c).

The reference ships no tests, golden vectors or sample data for this path, so
these are the build's own vectors: inputs come from the seeded generator or from
the seeded numpy streams below, and the files store the oracle's outputs (plus the generator KATs).
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    """name -> (src (3,n) f32, tgt (3,m) f32).  Shared with the tests (they re-create the inputs)."""
    out = {}
    # BASELINE config 1: 10k source / 1k target, unit cube, seed 0xC1
    out["c1"] = (O.synth_xyz(0xC1, 0, 10000), O.synth_xyz(0xC1, 1, 1000))
    # exact ties and duplicates: coarse lattice points, many at identical positions
    rng = np.random.default_rng(20261004)
    out["ties"] = (rng.integers(0, 8, size=(3, 4000)).astype(np.float32) / 8,
                   rng.integers(0, 16, size=(3, 400)).astype(np.float32) / 16)
    # queries outside the source bounding box (source in [0.25,0.75]^3, targets in [-1,2]^3)
    src = (0.25 + 0.5 * rng.random((3, 5000))).astype(np.float32)
    out["outside"] = (src, (-1.0 + 3.0 * rng.random((3, 300))).astype(np.float32))
    # k > N
    out["tiny"] = (rng.random((3, 5)).astype(np.float32), rng.random((3, 64)).astype(np.float32))
    # anisotropic / flat cloud: a thin slab (z extent 1e-3) -- surface-like
    flat = rng.random((3, 6000)).astype(np.float32)
    flat[2] *= 1e-3
    tflat = rng.random((3, 300)).astype(np.float32)
    tflat[2] *= 2e-3
    out["flat"] = (flat, tflat)
    return out


def main():
    store = {}
    for name, (src, tgt) in cases().items():
        ks = {"c1": (1, 8, 16, 32), "ties": (1, 8, 20), "outside": (8,), "tiny": (8, 32), "flat": (8,)}[name]
        for k in ks:
            sub = tgt[:, :256] if (name == "c1" and k > 8) else tgt
            idx, d2 = O.knn_bruteforce(src, sub, k)
            store["%s_k%d_idx" % (name, k)] = idx
            store["%s_k%d_d2" % (name, k)] = d2
    # blend vectors for c1/k8
    src, tgt = cases()["c1"]
    rgb, nrm = O.synth_rgb(0xC1, 10000), O.synth_nrm(0xC1, 10000)
    for mode in (0, 1):
        c, n = O.blend(store["c1_k8_idx"], store["c1_k8_d2"], rgb, nrm, mode)
        store["c1_k8_blend%d_rgb" % mode] = c
        store["c1_k8_blend%d_nrm" % mode] = n
    # PCA normals (build-defined) for c1 / k=16 (first 256 targets)
    pn, plan = O.pca_normals(store["c1_k16_idx"], src, nrm)
    store["c1_k16_pca_nrm"] = pn
    store["c1_k16_pca_planarity"] = plan
    # clustered generator KATs (seed 0xC5): first 8 sources / targets of a 1000 / 100 set
    store["kat_clustered_src"] = O.synth_xyz(0xC5, 0, 8, dist=1)
    store["kat_clustered_tgt"] = O.synth_xyz(0xC5, 1, 8, dist=1, n_total=1000, m_total=100)
    # generator KATs: first 8 values of each stream for seed 0xC1
    store["kat_src_xyz"] = O.synth_xyz(0xC1, 0, 8)
    store["kat_tgt_xyz"] = O.synth_xyz(0xC1, 1, 8)
    store["kat_rgb"] = O.synth_rgb(0xC1, 8)
    store["kat_nrm"] = O.synth_nrm(0xC1, 8)
    store["kat_splitmix"] = np.array([O.lib().pto_splitmix64(i) for i in range(4)], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "golden_knn.npz"), **store)
    print("wrote golden_knn.npz with", len(store), "arrays,", os.path.getsize(os.path.join(HERE, "golden_knn.npz")), "bytes")


if __name__ == "__main__":
    main()
