"""Throughput of the f32 traversal kernel by embedding width (the registry of src/embedding/models.rs has 384-, 768-, 1024-, 1536- and
3072-d models): rows x dims sized to ~15-30 GB, HNSW M = 32, efc = 128, 16 384 queries at ef = 64; algorithmic GB/s from the kernel's own counters."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import leann_rs_amd as la
L, chk = la.lib(), la._native.check
nq, k, ef = 16384, 10, 64
for d in [int(x) for x in (sys.argv[1:] or ["128", "256", "384", "512", "1024", "3072"])]:
    n = int(min(10_000_000, (24 << 30) // (d * 4)))
    X = la.DeviceArray((n, d), np.float32)
    chk(L.leann_synth_rows_device(0x5EED0001, d, d, min(64, d), 4096, 1.0, 0, 0, n, X.ptr, None)); la.sync()
    Q = la.DeviceArray((nq, d), np.float32)
    chk(L.leann_synth_rows_device(0x5EED0001, d, d, min(64, d), 4096, 1.0, 1, 0, nq, Q.ptr, None)); la.sync()
    t0 = time.time()
    s = la.BackendSearcher.build_device(0, X.ptr, n, d, d, 32, 128)
    tb = time.time() - t0
    ok, od, oc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    st = la.DeviceArray((nq, 4), np.uint32)
    s.search_batch_device(Q.ptr, nq, k, ef, ok.ptr, od.ptr, oc.ptr, st.ptr, None); la.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        s.search_batch_device(Q.ptr, nq, k, ef, ok.ptr, od.ptr, oc.ptr, st.ptr, None)
    la.sync()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    h = st.to_host().astype(np.int64)
    by = (h[:, 0].sum() * d * 4 + h[:, 1].sum() * 64 * 4 + h[:, 2].sum() * 32 * 4)
    print(f"d={d:5d} n={n:9d} build {tb:5.1f} s: {ms:7.3f} ms per 16384 queries = {nq / ms * 1e3 / 1e6:.3f} M q/s, {by / ms / 1e9:.2f} TB/s algorithmic ({h[:, 0].mean():.0f} evals/query)", flush=True)
    s.close(); del X, Q
