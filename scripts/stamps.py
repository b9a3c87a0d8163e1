"""Reads the per-phase cycle sums written by the LEANN_STAMPS diagnostic build (scripts/stamps.sh)."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leann_rs_amd as la
L, chk = la.lib(), la._native.check
n, d, ef = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000, 768, int(sys.argv[2]) if len(sys.argv) > 2 else 64
feat = len(sys.argv) > 3 and sys.argv[3] == "feat"
SEED = 0x5EED0001
if feat:
    F = la.DeviceArray((n, 256), np.uint16); W = la.DeviceArray((256, d), np.uint16)
    chk(L.leann_synth_features_device(SEED, 256, 64, 4096, 1.0, 0, 0, n, F.ptr, None)); chk(L.leann_synth_weights_device(SEED, 256, d, W.ptr, None)); la.sync()
    r = C.c_void_p(); chk(L.leann_recompute_create(F.ptr, n, 256, W.ptr, d, 0, 0, C.byref(r)))
    hb = C.c_void_p(); chk(L.leann_recompute_build_index(r, 0, 32, 128, C.byref(hb))); s = la.BackendSearcher(hb, 0)
else:
    X = la.DeviceArray((n, d), np.float32)
    chk(L.leann_synth_rows_device(SEED, d, d, 64, 4096, 1.0, 0, 0, n, X.ptr, None)); la.sync()
    s = la.BackendSearcher.build_device(0, X.ptr, n, d, d, 32, 128)
for nq in (16384, 64):
    Q = la.DeviceArray((nq, d), np.float32)
    chk(L.leann_synth_rows_device(SEED, d, d, 64, 4096, 1.0, 1, 0, nq, Q.ptr, None)); la.sync()
    k = 10
    ok, od, oc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    stamps = la.DeviceArray((nq, 8), np.uint64)
    os.environ["LEANN_STAMP_BUF"] = str(stamps.ptr)
    la.lib().leann_debug_reload_env()  # the library reads its knobs once
    for _ in range(2):
        s.search_batch_device(Q.ptr, nq, k, ef, ok.ptr, od.ptr, oc.ptr, None, None); la.sync()
    st = stamps.to_host().astype(np.float64)
    hops = st[:, 6].sum()
    # throughput form (batch 16384): B at the head of the hop; latency form (batch 64, 16 waves): B runs inside the E/D phase of the hop before
    names = ["B adjacency+visited (wave0; latency form: 0)", "wait B1 (latency form: 0)", "C rows+dist", "wait B2", "E next candidate (+B) / D merge", "wait B3"]
    tot = st[:, :6].sum()
    print(f"nq={nq} ef={ef} feat={feat}: {hops/nq:.0f} hops/query, {tot/hops:.0f} cycles per hop (wave 0 view)")
    for i, nm in enumerate(names):
        print(f"   {nm:30s} {st[:, i].sum()/hops:8.0f} cycles  {st[:, i].sum()/tot*100:5.1f} %")
    print(f"   next candidate taken from the old beam (not from the hop's new keys): {st[:, 7].sum()/hops*100:.1f} % of the hops")
