"""Single-query latency of the two filtered-search paths (graph walk with the filter inside vs exact scan of the allowed rows) through
the host-pointer C ABI, the way the C++ IndexSearcher calls them.  Usage (GPU box): python scripts/filter_latency.py [rows] [dims]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import leann_rs_amd as la

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
dev = torch.device("cuda", 0)
X = torch.empty((rows, d), dtype=torch.float32, device=dev)
Q = torch.empty((64, d), dtype=torch.float32, device=dev)
L, chk = la.lib(), la._native.check
chk(L.leann_synth_rows_device(0x5EED0001, d, d, 64, 4096, 1.0, 0, 0, rows, X.data_ptr(), None))
chk(L.leann_synth_rows_device(0x5EED0001, d, d, 64, 4096, 1.0, 1, 0, 64, Q.data_ptr(), None))
torch.cuda.synchronize()
t0 = time.time()
s = la.BackendSearcher.build_device(la.BackendType.Hnsw, X.data_ptr(), rows, d, d, 32, 128)
print(f"index over {rows} x {d} built in {time.time() - t0:.1f}s", flush=True)
Qh = Q.cpu().numpy()
rng = np.random.default_rng(5)
for sel in (0.1, 0.03, 0.01, 0.001):
    bm = np.packbits(rng.random(rows) < sel, bitorder="little")
    ef = int(min(1024, max(64, np.ceil(8 * 10 / (20 * sel)))))
    out = {}
    for name, fn in (("walk", lambda q: s.search_filtered_batch(q, 10, ef, bm)), ("exact", lambda q: s.search_filtered_exact_batch(q, 10, bm))):
        for i in range(3):
            fn(Qh[i:i + 1])
        t0 = time.perf_counter()
        for i in range(20):
            fn(Qh[i:i + 1])
        out[name] = (time.perf_counter() - t0) / 20 * 1e3
    f = s.register_filter(bm)  # uploaded + compacted once: what the C++ IndexSearcher does per distinct filter
    for name, mode in (("registered/walk", "walk"), ("registered/exact", "exact"), ("registered/auto", "auto")):
        for i in range(3):
            s.search_filter_batch(Qh[i:i + 1], 10, ef, f, mode)
        t0 = time.perf_counter()
        for i in range(20):
            s.search_filter_batch(Qh[i:i + 1], 10, ef, f, mode)
        out[name] = (time.perf_counter() - t0) / 20 * 1e3
    f.close()
    ek = s.search_filtered_exact_batch(Qh, 10, bm)[0]
    wk = s.search_filtered_batch(Qh, 10, ef, bm)[0]
    rec = np.mean([len(set(ek[i].tolist()) & set(wk[i].tolist())) / 10 for i in range(64)])
    print(f"allowed {sel:6.3f} ({int(sel * rows)} rows): walk ef={ef}: {out['walk']:.3f} ms/query (recall {rec:.3f}), exact: {out['exact']:.3f} ms/query; registered filter: walk {out['registered/walk']:.3f}, exact {out['registered/exact']:.3f}, auto {out['registered/auto']:.3f} ms/query", flush=True)
s.close()
