"""Ad-hoc scale exploration on the GPU box: build time, recall, QPS.  Not part of the test suite."""
import argparse, sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leann_rs_amd as la

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--d", type=int, default=768)
ap.add_argument("--M", type=int, default=32)
ap.add_argument("--efc", type=int, default=128)
ap.add_argument("--ef", type=str, default="32,64,128")
ap.add_argument("--nq", type=int, default=16384)
ap.add_argument("--ngt", type=int, default=1000)
ap.add_argument("--r", type=int, default=64)
ap.add_argument("--clusters", type=int, default=4096)
ap.add_argument("--sigma", type=float, default=1.0)
ap.add_argument("--backend", type=int, default=0)
a = ap.parse_args()
L = la.lib(); chk = la._native.check
n, d = a.n, a.d
ld = (d + 3) // 4 * 4
SEED = 0x5EED0001
t = time.time()
X = la.DeviceArray((n, ld), np.float32)
chk(L.leann_synth_rows_device(SEED, d, ld, a.r, a.clusters, a.sigma, 0, 0, n, X.ptr, None)); la.sync()
print(f"gen corpus {n}x{d}: {time.time()-t:.2f}s", flush=True)
Q = la.DeviceArray((a.nq, ld), np.float32)
chk(L.leann_synth_rows_device(SEED, d, ld, a.r, a.clusters, a.sigma, 1, 0, a.nq, Q.ptr, None)); la.sync()
assert ld == d
t = time.time()
s = la.BackendSearcher.build_device(a.backend, X.ptr, n, d, ld, a.M, a.efc)
la.sync(); bt = time.time() - t
gi = s.graph_info()
print(f"build: {bt:.1f}s  ({n/bt:.0f} pts/s) max_level={gi['max_level']} entry={gi['entry']}", flush=True)
# ground truth
k = 10
t = time.time()
gk = la.DeviceArray((a.ngt, k), np.uint64); gs = la.DeviceArray((a.ngt, k), np.float32); gc = la.DeviceArray(a.ngt, np.uint32)
chk(L.leann_scan_topk_device(X.ptr, n, d, ld, Q.ptr, a.ngt, k, None, 0, gk.ptr, gs.ptr, gc.ptr, None)); la.sync()
print(f"ground truth ({a.ngt} q): {time.time()-t:.2f}s", flush=True)
truth = gk.to_host()
ok = la.DeviceArray((a.nq, k), np.uint64); od = la.DeviceArray((a.nq, k), np.float32); oc = la.DeviceArray(a.nq, np.uint32)
st = la.DeviceArray((a.nq, 4), np.uint32)
for ef in [int(x) for x in a.ef.split(",")]:
    s.search_batch_device(Q.ptr, a.nq, k, ef, ok.ptr, od.ptr, oc.ptr, st.ptr, None); la.sync()
    reps = 3
    t = time.time()
    for _ in range(reps):
        s.search_batch_device(Q.ptr, a.nq, k, ef, ok.ptr, od.ptr, oc.ptr, st.ptr, None)
    la.sync(); dt = (time.time() - t) / reps
    keys = ok.to_host(); stats = st.to_host().astype(np.int64)
    rec = np.mean([len(set(keys[i].tolist()) & set(truth[i].tolist())) / k for i in range(a.ngt)])
    ev, h0, hu, ovf = stats.sum(0)
    pe = np.percentile(stats[:, 0], [50, 90, 99, 99.9, 100]).astype(int)
    by = ev * d * 4 + h0 * gi['M0'] * 4 + hu * gi['M'] * 4
    print(f"ef={ef}: {dt*1e3:.2f} ms / {a.nq} q -> {a.nq/dt:.0f} QPS  recall@10={rec:.4f}  evals/q={ev/a.nq:.0f} hops/q={h0/a.nq:.0f}+{hu/a.nq:.0f} "
          f"evals pct50/90/99/99.9/max={pe.tolist()} ovf={ovf} bytes/q={by/a.nq/1e6:.2f}MB  {by/dt/1e12:.3f} TB/s = {by/dt/8e12*100:.1f}% of 8 TB/s", flush=True)
deg = (s.graph_export()['adj0'] != 0xFFFFFFFF).sum(1)
print("level-0 degree: mean %.1f min %d max %d" % (deg.mean(), deg.min(), deg.max()))
