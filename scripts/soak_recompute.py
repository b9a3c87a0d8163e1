"""Randomised soak of the exhaustive recompute search (feature-stationary kernel, multi-chunk candidate emission, multi-tile query
batches, allow masks) against the oracle's literal embed -> dot -> sort.  Usage (GPU box): python scripts/soak_recompute.py [n_configs] [seed]"""
import ctypes as C, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po
import leann_rs_amd as la

L, chk = la.lib(), la._native.check
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
SEED, h, bad = 0x5EED0001, 256, 0
for c in range(n_cfg):
    n = int(rng.integers(66000, 140000)); d = int(rng.choice([384, 512, 768])); nq = int(rng.choice([1, 33, 64, 130, 257]))
    k = int(rng.integers(1, 40)); masked = bool(rng.integers(0, 2))
    F = po.synth_features(SEED + c, h, 512, 1.0, 0, 0, n, r_int=48)
    W = po.synth_weights(SEED + c, h, d)
    with ThreadPoolExecutor(8) as ex:
        parts = list(ex.map(lambda b: po.recompute_encode(F[b[0]:b[1]], W), [(n * t // 8, n * (t + 1) // 8) for t in range(8)]))
    E = np.concatenate(parts)
    Q = po.recompute_encode(po.synth_features(SEED + c, h, 512, 1.0, 1, 0, nq, r_int=48), W)
    mask = None
    if masked:
        mask = np.packbits(rng.random(n) < 0.3, bitorder="little")
    dF, dW, dQ = la.DeviceArray.from_host(F), la.DeviceArray.from_host(W), la.DeviceArray.from_host(Q)
    dM = la.DeviceArray.from_host(mask) if masked else None
    r = C.c_void_p()
    chk(L.leann_recompute_create(dF.ptr, n, h, dW.ptr, d, 0, 0, C.byref(r)))
    dk, ds, dc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    chk(L.leann_recompute_search_batch_device(r, dQ.ptr, nq, k, dM.ptr if masked else None, dk.ptr, ds.ptr, dc.ptr, None))
    la.sync()
    gk, gs = dk.to_host().astype(np.int64), ds.to_host()
    worst, miss = 0.0, 0
    for i in rng.choice(nq, size=min(nq, 6), replace=False):
        k0, s0 = po.scan_topk(E, Q[i], k, mode=0, allow_mask=mask)
        worst = max(worst, float(np.abs(gs[i] - s0).max()))
        for j in range(k):
            if gk[i, j] != k0[j] and abs(float(E[gk[i, j]] @ Q[i]) - float(s0[j])) > 2e-5:
                miss += 1
        if masked:
            assert ((mask[gk[i] >> 3] >> (gk[i] & 7)) & 1).all()
    ok = worst <= 1e-5 and miss == 0
    print(f"cfg {c}: n={n} d={d} nq={nq} k={k} mask={masked}: max |score - oracle| = {worst:.2e}, ids off outside near-ties: {miss} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
    L.leann_recompute_close(r)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
