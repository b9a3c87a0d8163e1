"""Randomised soak of sharded (composite) handles — not part of the test suite: random (rows, dims, shard count, top_k, complexity, filter
density, filter mode) configurations on one device; the composite handle's answers against ONE unsharded handle over the same rows for
everything exact (exact filtered search, registered filters in exact / auto-exact mode), and against the per-shard walks merged on the
host for the graph walks (plain, in-traversal bitmap, registered filter in walk mode).  Usage (GPU box): python scripts/soak_shard.py [n] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po
import leann_rs_amd as la

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1357)
U64MAX = np.iinfo(np.uint64).max
bad = 0


def same_up_to_ties(a, b):
    """(keys, dists, counts) equal; entries of equal distance may come in another order (1 - score rounds)"""
    if not ((a[1] == b[1]).all() and (a[2] == b[2]).all()):
        return False
    return all((a[0][q][np.lexsort((a[0][q], a[1][q]))] == b[0][q][np.lexsort((b[0][q], b[1][q]))]).all() for q in range(len(a[0])))


for c in range(n_cfg):
    n = int(rng.integers(700, 12000)); d = int(rng.choice([64, 128, 256, 768])); G = int(rng.choice([2, 3, 4, 8])); M = int(rng.choice([8, 16]))
    k = int(rng.integers(1, 40)); ef = int(rng.integers(k, 120)); nq = int(rng.choice([1, 9, 64])); dens = float(rng.choice([0.003, 0.02, 0.3, 0.8]))
    X = po.gen_rows(0x5EED0001 + c, d, 32, 97, 0.8, 0, 0, n); Q = po.gen_rows(0x5EED0001 + c, d, 32, 97, 0.8, 1, 0, nq)
    lows = [0] + [((n * g) // G) & ~63 for g in range(1, G)] + [n]
    if any(lows[g + 1] <= lows[g] for g in range(G)):
        continue
    parts = [la.DeviceArray.from_host(X[lows[g]:lows[g + 1]]) for g in range(G)]
    s = la.ShardedIndex.build_device(0, [p.ptr for p in parts], [lows[g + 1] - lows[g] for g in range(G)], d, d, M, 48, [0] * G, keep=parts).as_backend()
    dX = la.DeviceArray.from_host(X)
    one = la.BackendSearcher.build_device(0, dX.ptr, n, d, d, M, 48)
    allow = np.packbits(rng.random(n) < dens, bitorder="little")
    ok = True
    # exact paths == the unsharded handle
    ok &= same_up_to_ties(s.search_filtered_exact_batch(Q, k, allow), one.search_filtered_exact_batch(Q, k, allow))
    f, f1 = s.register_filter(allow), one.register_filter(allow)
    ok &= f.count() == f1.count()
    ok &= same_up_to_ties(s.search_filter_batch(Q, k, ef, f, mode="exact"), one.search_filter_batch(Q, k, ef, f1, mode="exact"))
    ok &= same_up_to_ties(s.search_filter_batch(Q, k, ef, f, mode="auto"), one.search_filter_batch(Q, k, ef, f1, mode="auto")) \
        if f.count() <= max(0.05 * n, 65536) and nq <= 64 else True  # (auto == exact at these sizes for both)
    # walks == per-shard walks merged by (dist, key) on the host
    def merged(fn):
        ks, ds = [], []
        for g in range(G):
            kk, dd, cc = fn(s.shard(g), lows[g], lows[g + 1])
            ks.append(kk); ds.append(dd)
        out_k = np.full((nq, k), U64MAX, np.uint64); out_d = np.full((nq, k), np.inf, np.float32); out_c = np.zeros(nq, np.uint32)
        for q in range(nq):
            kq = np.concatenate([a[q] for a in ks]); dq = np.concatenate([a[q] for a in ds]); v = kq != U64MAX
            o = np.lexsort((kq[v], dq[v]))[:k]
            out_k[q, :len(o)], out_d[q, :len(o)], out_c[q] = kq[v][o], dq[v][o], len(o)
        return out_k, out_d, out_c
    w = merged(lambda sh, lo, hi: sh.search_batch(Q, k, ef))
    g_ = s.search_batch(Q, k, ef)
    ok &= all((a == b).all() for a, b in zip(g_, w))
    wf = merged(lambda sh, lo, hi: sh.search_filtered_batch(Q, k, ef, allow[lo // 8:][: (hi - lo + 7) // 8]))
    ok &= all((a == b).all() for a, b in zip(s.search_filtered_batch(Q, k, ef, allow), wf))
    ok &= all((a == b).all() for a, b in zip(s.search_filter_batch(Q, k, ef, f, mode="walk"), wf))
    f.close(); f1.close(); s.close(); one.close()
    bad += 0 if ok else 1
    print(f"cfg {c}: n={n} d={d} G={G} M={M} k={k} ef={ef} nq={nq} density={dens}: {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"mismatches: {bad}")
sys.exit(1 if bad else 0)
