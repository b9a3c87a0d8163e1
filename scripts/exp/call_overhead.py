"""Host-side cost of one leann_backend_search call: a tiny index (the kernel walks a few hops), so what is timed is staging + launches + sync."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import leann_rs_amd as la
rng = np.random.default_rng(0)
for d in (128, 768):
    X = rng.standard_normal((2000, d)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
    dX = la.DeviceArray.from_host(X)
    s = la.BackendSearcher.build_device(0, dX.ptr, X.shape[0], d, d, 8, 32)
    q = X[:1].copy()
    for env in ("0", "1"):
        if env == "1": os.environ["LEANN_DEBUG_NO_ZERO_COPY"] = "1"
        else: os.environ.pop("LEANN_DEBUG_NO_ZERO_COPY", None)
        la.lib().leann_debug_reload_env()  # the library reads its knobs once
        for _ in range(200): s.search_batch(q, 5, 8)
        t = []
        for _ in range(3000):
            t0 = time.perf_counter(); s.search_batch(q, 5, 8); t.append(time.perf_counter() - t0)
        t = np.array(t) * 1e6
        print(f"d={d} {'staged copies' if env == '1' else 'zero-copy   '}: p50 {np.percentile(t, 50):.1f} us  p99 {np.percentile(t, 99):.1f} us  (incl. the ctypes call)")
    s.close()
