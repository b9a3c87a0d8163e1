import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import leann_rs_amd as la
import pyoracle as po
from util import synth
G = 8
n, d, nq, k, ef, M = 8192 + 640, 128, 40, 10, 48, 12
X = synth(po, n, d); Q = synth(po, nq, d, stream=1)
lows = [0] + [((n * g) // G) & ~63 for g in range(1, G)] + [n]
parts = [la.DeviceArray.from_host(X[lows[g]:lows[g + 1]]) for g in range(G)]
s = la.ShardedIndex.build_device(0, [p.ptr for p in parts], [lows[g + 1] - lows[g] for g in range(G)], d, d, M, 48, [0] * G, keep=parts).as_backend()
dX = la.DeviceArray.from_host(X)
one = la.BackendSearcher.build_device(0, dX.ptr, n, d, d, M, 48)
rng = np.random.default_rng(100 + G)
sparse = np.packbits(rng.random(n) < 0.01, bitorder="little")
dense = np.packbits(rng.random(n) < 0.3, bitorder="little")
K = 600
bk, bd, bc = one.search_filtered_exact_batch(Q[:6], K, dense)
gk, gd, gc = s.search_filtered_exact_batch(Q[:6], K, dense)
print("unsharded counts", bc, "sharded counts", gc)
print("keys equal", (bk == gk).all(), "dists equal", (bd == gd).all())
allk, alld = [], []
for g in range(G):
    sg = s.shard(g)
    sl = dense[lows[g] // 8:][: (lows[g + 1] - lows[g] + 7) // 8]
    pk, pd, pc = sg.search_filtered_exact_batch(Q[:6], K, sl)
    print("shard", g, "rows", lows[g + 1] - lows[g], "allowed", int(np.unpackbits(sl, bitorder='little')[:lows[g+1]-lows[g]].sum()), "counts", pc)
    allk.append(pk); alld.append(pd)
for q in range(6):
    ks = np.concatenate([a[q] for a in allk]); ds = np.concatenate([a[q] for a in alld])
    v = ks != np.iinfo(np.uint64).max
    o = np.lexsort((ks[v], ds[v]))
    mk = ks[v][o][:K]; md = ds[v][o][:K]
    print(q, "host-merge == unsharded:", (mk == bk[q][:len(mk)]).all() and len(mk) == bc[q], " == sharded:", (mk == gk[q][:len(mk)]).all(), "first diff sharded", np.nonzero(mk != gk[q][:len(mk)])[0][:3])
