"""Randomised parity soak (not part of the test suite): random (n, dims, M, ef, k, batch, filter) configurations, HIP path vs oracle,
bit-exact ids / distances / counters.  Usage (GPU box): python scripts/soak_parity.py [n_configs] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po
import leann_rs_amd as la

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
bad = 0
for c in range(n_cfg):
    n = int(rng.integers(50, 6000)); d = int(rng.choice([16, 64, 100, 128, 260, 384, 768, 1000, 1536, 2500, 3072, 4096])); M = int(rng.choice([2, 4, 8, 16, 32]))
    ef = int(rng.integers(1, 300)); k = int(rng.integers(1, min(ef, 64) + 1)); nq = int(rng.choice([1, 7, 64, 400, 600, 700]))
    r = int(rng.choice([0, 8, 64])); vam = bool(rng.integers(0, 2))
    X = po.gen_rows(0x5EED0001 + c, d, min(r, d), 97, 0.8, 0, 0, n)
    Q = po.gen_rows(0x5EED0001 + c, d, min(r, d), 97, 0.8, 1, 0, nq)
    if vam:
        G = po.Graph.build_vamana(X, R=max(M, 4), L=max(2 * M, 8), alpha=1.2)
        lv, uo, a0, aU = G.export()
        s = la.BackendSearcher.from_arrays(la.BackendType.DiskAnn, X, max(M, 4), max(M, 4), 0, G.entry, lv, uo, a0, aU)
    else:
        G = po.Graph.build_hnsw(X, M=M, efc=max(2 * M, 16))
        lv, uo, a0, aU = G.export()
        s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, M, 2 * M, G.max_level, G.entry, lv, uo, a0, aU)
    algo = 1 if vam else 0
    ok, od, oc, ost = G.search_batch(Q, k, ef, algo, nthreads=8)
    s.stats(reset=True)
    gk, gd, gc = s.search_batch(Q, k, ef)
    st = s.stats()
    same = (gk == ok).all() and (gd.view(np.uint32) == od.view(np.uint32)).all() and (gc == oc).all() and st["n_dist_evals"] == int(ost[:, 0].sum())
    bm = np.packbits(rng.random(n) < rng.choice([0.5, 0.1, 0.02]), bitorder="little")
    fk, fd, fc, _ = G.search_filtered_batch(Q, k, ef, bm, algo, nthreads=8)
    hk, hd, hc = s.search_filtered_batch(Q, k, ef, bm)
    same_f = (hk == fk).all() and (hd.view(np.uint32) == fd.view(np.uint32)).all() and (hc == fc).all()
    # exact filtered search (compacted allowed rows + f32 MFMA scan) vs the oracle's sequential-fmaf scan with the early filter
    xk, xd, xc = s.search_filtered_exact_batch(Q, k, bm)
    same_x = True
    for i in range(min(nq, 40)):
        rk, rs = po.scan_topk(X, Q[i], k, mode=1, allow_mask=bm)
        same_x &= xc[i] == len(rk) and (xk[i, : len(rk)] == rk).all() and (xd[i, : len(rk)].view(np.uint32) == (np.float32(1.0) - rs).view(np.uint32)).all()
    print(f"cfg {c:2d}: n={n} d={d} M={M} ef={ef} k={k} nq={nq} r={r} {'vamana' if vam else 'hnsw'}: search {'ok' if same else 'MISMATCH'}, filtered {'ok' if same_f else 'MISMATCH'}, exact-filtered {'ok' if same_x else 'MISMATCH'}", flush=True)
    bad += (not same) + (not same_f) + (not same_x)
    s.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
