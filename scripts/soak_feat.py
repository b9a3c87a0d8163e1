"""Randomised parity soak of the recompute-on GRAPH search (no stored vectors; not part of the test suite): random (n, feature width,
dims, M, ef, k, batch, filter density) configurations, HIP traversal vs the oracle walking the same graph and the same feature bytes —
bit-exact ids / distances / counters, plain and filtered, in both forms of the hop loop (batches <= 512: 16 waves per query, larger:
4) and for rows of 256 features (four rows per wave instruction) as well as other widths (one row per wave load).
Usage (GPU box): python scripts/soak_feat.py [n_configs] [seed]"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po
import leann_rs_amd as la

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4321)
Lc, chk = la.lib(), la._native.check
bad = 0
for c in range(n_cfg):
    n = int(rng.integers(300, 30000)); h = int(rng.choice([64, 128, 256, 256, 256])); d = int(rng.choice([128, 384, 768]))
    M = int(rng.choice([4, 8, 16, 32])); ef = int(rng.integers(1, 200)); k = int(rng.integers(1, min(ef, 48) + 1))
    nq = int(rng.choice([1, 7, 64, 513, 700, 1300])); seed = 0x5EED0001 + c
    F = po.synth_features(seed, h, min(64, h), 1.0, 0, 0, n)
    W = po.synth_weights(seed, h, d)
    Q = po.recompute_encode(po.synth_features(seed, h, min(64, h), 1.0, 1, 0, nq), W)
    dF, dW = la.DeviceArray.from_host(F), la.DeviceArray.from_host(W)
    r = C.c_void_p(); chk(Lc.leann_recompute_create(dF.ptr, n, h, dW.ptr, d, 0, 0, C.byref(r)))
    hb = C.c_void_p(); chk(Lc.leann_recompute_build_index(r, 0, M, max(2 * M, 32), C.byref(hb)))
    s = la.BackendSearcher(hb, la.BackendType.Hnsw)
    fh, rb = C.c_uint32(0), C.c_uint32(0)
    chk(Lc.leann_backend_feature_rows_export(hb, C.byref(fh), C.byref(rb), None))
    rows = np.zeros((n, rb.value), np.uint8)
    chk(Lc.leann_backend_feature_rows_export(hb, None, None, rows.ctypes.data))
    g = s.graph_export()
    Gr = po.Graph.from_arrays(np.zeros((n, 1), np.float32), M, 2 * M, g["max_level"], g["entry"], g["levels"], g["upper_off"], g["adj0"], g["adjU"])
    Gr.set_features(rows, fh.value, rb.value)
    PQ = po.project_queries(W, Q, fh.value)
    ok, od, oc, ost = Gr.search_batch(PQ, k, ef, 0, 8)
    s.stats(reset=True)
    gk, gd, gc = s.search_batch(Q, k, ef)
    st = s.stats()
    same = (gk == ok).all() and (gd.view(np.uint32) == od.view(np.uint32)).all() and (gc == oc).all() and st["n_dist_evals"] == int(ost[:, 0].sum())
    bm = np.packbits(rng.random(n) < rng.choice([0.5, 0.1, 0.02]), bitorder="little")
    fk, fd, fc, _ = Gr.search_filtered_batch(PQ, k, ef, bm, 0, 8)
    hk, hd, hc = s.search_filtered_batch(Q, k, ef, bm)
    same_f = (hk == fk).all() and (hd.view(np.uint32) == fd.view(np.uint32)).all() and (hc == fc).all()
    print(f"cfg {c:2d}: n={n} h={h} d={d} M={M} ef={ef} k={k} nq={nq}: search {'ok' if same else 'MISMATCH'}, filtered {'ok' if same_f else 'MISMATCH'}", flush=True)
    bad += (not same) + (not same_f)
    s.close()
    Lc.leann_recompute_close(r)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
