"""Is the 2-4 % run-to-run spread of the headline kernel (14.05 / 14.35 / 14.65 ms per 16 384 queries, stable within a process) a property of
WHERE the rows landed?  One process, the same 10M x 768 rows in two allocations (the second a copy), the same graph (the builder is
deterministic) in two handles; the query kernel timed alternately on both."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import leann_rs_amd as la
L, chk = la.lib(), la._native.check
n, d, nq, k, ef = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 768, 16384, 10, 56
SEED = 0x5EED0001
X = la.DeviceArray((n, d), np.float32)
chk(L.leann_synth_rows_device(SEED, d, d, 64, 4096, 1.0, 0, 0, n, X.ptr, None)); la.sync()
Q = la.DeviceArray((nq, d), np.float32)
chk(L.leann_synth_rows_device(SEED, d, d, 64, 4096, 1.0, 1, 0, nq, Q.ptr, None)); la.sync()
hs = []
for copy in (0, 1, 1):
    t0 = time.time()
    hs.append(la.BackendSearcher.build_device(0, X.ptr, n, d, d, 32, 200, take_copy=bool(copy)))
    print(f"handle {len(hs)} built in {time.time() - t0:.1f} s (rows {'copied into an allocation of its own' if copy else 'borrowed'})", flush=True)
ok, od, oc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
ref = None
for rep in range(3):
    for i, s in enumerate(hs):
        s.search_batch_device(Q.ptr, nq, k, ef, ok.ptr, od.ptr, oc.ptr, None, None); la.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            s.search_batch_device(Q.ptr, nq, k, ef, ok.ptr, od.ptr, oc.ptr, None, None)
        la.sync()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        keys = ok.to_host()
        if ref is None: ref = keys.copy()
        print(f"rep {rep} handle {i + 1}: {ms:.3f} ms per launch  ({nq / ms * 1e3:.0f} queries/s)  same answers as handle 1: {(keys == ref).all()}", flush=True)
