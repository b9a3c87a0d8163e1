"""Waves per query by batch size: time per launch of nq queries with 4 / 8 / 16 waves per query (LEANN_DEBUG_NW), 10M x 768, ef = 56."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import leann_rs_amd as la
L, chk = la.lib(), la._native.check
n, d, k, ef = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 768, 10, 56
X = la.DeviceArray((n, d), np.float32)
chk(L.leann_synth_rows_device(0x5EED0001, d, d, 64, 4096, 1.0, 0, 0, n, X.ptr, None)); la.sync()
NQ = 8192
Q = la.DeviceArray((NQ, d), np.float32)
chk(L.leann_synth_rows_device(0x5EED0001, d, d, 64, 4096, 1.0, 1, 0, NQ, Q.ptr, None)); la.sync()
s = la.BackendSearcher.build_device(0, X.ptr, n, d, d, 32, 200)
ok, od, oc = la.DeviceArray((NQ, k), np.uint64), la.DeviceArray((NQ, k), np.float32), la.DeviceArray(NQ, np.uint32)
for nq in (64, 256, 512, 768, 1024, 1536, 2048, 3072, 4096, 8192):
    line = f"nq={nq:5d}:"
    for nw in (4, 8, 16):
        os.environ["LEANN_DEBUG_NW"] = str(nw); L.leann_debug_reload_env()
        s.search_batch_device(Q.ptr, nq, k, ef, ok.ptr, od.ptr, oc.ptr, None, None); la.sync()
        reps = 20 if nq <= 1024 else 8
        t0 = time.perf_counter()
        for _ in range(reps):
            s.search_batch_device(Q.ptr, nq, k, ef, ok.ptr, od.ptr, oc.ptr, None, None)
        la.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        line += f"  {nw:2d} waves {ms:7.3f} ms ({nq / ms:7.1f} k q/s)"
    print(line, flush=True)
