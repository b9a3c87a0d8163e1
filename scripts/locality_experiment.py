"""Experiment: does a locality-preserving row order (rows of a cluster adjacent in memory) speed the traversal up?  The synthetic corpus
assigns clusters by a hash of the position, so neighbours sit at random addresses; here the same rows are permuted cluster-major before the
build.  Usage (GPU box): python scripts/locality_experiment.py [n]"""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import leann_rs_amd as la
L, chk = la.lib(), la._native.check
n, d, M, efc, k, NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000, 768, 32, 200, 10, 16384
SEED, NCL = 0x5EED0001, 4096
dev = torch.device("cuda", 0)
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
    return x ^ (x >> np.uint64(31))


with np.errstate(over="ignore"):
    i = np.arange(n, dtype=np.uint64)
    seed_a = np.uint64(SEED ^ 0x4153534700000000)
    h = mix64(mix64(seed_a ^ (np.uint64(0) * np.uint64(0xD1342543DE82EF95))) ^ (i * np.uint64(0xA24BAED4963EE407)))
cluster = (h % np.uint64(NCL)).astype(np.int64)
X = torch.empty((n, d), dtype=torch.float32, device=dev)
chk(L.leann_synth_rows_device(SEED, d, d, 64, NCL, 1.0, 0, 0, n, X.data_ptr(), None))
Q = torch.empty((NQ, d), dtype=torch.float32, device=dev)
chk(L.leann_synth_rows_device(SEED, d, d, 64, NCL, 1.0, 1, 0, NQ, Q.data_ptr(), None))
torch.cuda.synchronize()
gt_k = torch.empty((1000, k), dtype=torch.int64, device=dev); gt_s = torch.empty((1000, k), dtype=torch.float32, device=dev); gt_c = torch.empty((1000,), dtype=torch.int32, device=dev)
keys = torch.empty((NQ, k), dtype=torch.int64, device=dev); dists = torch.empty((NQ, k), dtype=torch.float32, device=dev); cnt = torch.empty((NQ,), dtype=torch.int32, device=dev)
for name in ("hashed order (baseline)", "cluster-major order"):
    if name.startswith("cluster"):
        perm = torch.from_numpy(np.argsort(cluster, kind="stable")).to(dev)
        X = X.index_select(0, perm).contiguous()
        del perm
        torch.cuda.synchronize()
    chk(L.leann_scan_topk_device(X.data_ptr(), n, d, d, Q.data_ptr(), 1000, k, None, 0, gt_k.data_ptr(), gt_s.data_ptr(), gt_c.data_ptr(), None))
    torch.cuda.synchronize()
    truth = gt_k.cpu().numpy()
    t = time.time()
    s = la.BackendSearcher.build_device(0, X.data_ptr(), n, d, d, M, efc)
    torch.cuda.synchronize()
    print(f"{name}: build {time.time() - t:.1f} s", flush=True)
    for ef in (48, 56, 64):
        for _ in range(2):
            s.search_batch_device(Q.data_ptr(), NQ, k, ef, keys.data_ptr(), dists.data_ptr(), cnt.data_ptr(), None, None)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            s.search_batch_device(Q.data_ptr(), NQ, k, ef, keys.data_ptr(), dists.data_ptr(), cnt.data_ptr(), None, None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 10
        got = keys[:1000].cpu().numpy()
        rec = np.mean([len(set(got[i].tolist()) & set(truth[i].tolist())) / k for i in range(1000)])
        print(f"  ef={ef}: {dt * 1e3:.2f} ms / {NQ} queries -> {NQ / dt / 1e6:.3f} M QPS, recall@10 {rec:.4f}", flush=True)
    s.close()
