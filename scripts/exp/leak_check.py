"""Open / search / close cycles: device memory, pinned blocks and dispatcher threads must all go away with the handle."""
import os, sys, threading
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import leann_rs_amd as la
rng = np.random.default_rng(0)
X = rng.standard_normal((100000, 128)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
dX = la.DeviceArray.from_host(X)
def cycle(what):
    s = la.BackendSearcher.build_device(0, dX.ptr, X.shape[0], 128, 128, 16, 64)
    if "staged" in what: s.search_batch(X[:700], 10, 64)
    if "zero" in what: s.search_batch(X[:3], 10, 64)
    if "device" in what:
        k = la.DeviceArray((64, 10), np.uint64); d = la.DeviceArray((64, 10), np.float32); c = la.DeviceArray(64, np.uint32)
        s.search_batch_device(dX.ptr, 64, 10, 64, k.ptr, d.ptr, c.ptr, None, None); la.sync()
    if "threads" in what:
        th = [threading.Thread(target=lambda i=i: [s.search(X[i], 5, 32) for _ in range(20)]) for i in range(16)]
        [t.start() for t in th]; [t.join() for t in th]
    s.close()
for what in ("build only", "device", "staged", "zero", "threads", "staged zero threads"):
    cycle(what); torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(); n0 = len(os.listdir("/proc/self/task"))
    for _ in range(20): cycle(what)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    print(f"{what:22s}: 20 more cycles cost {(free0 - free1) / 2**20:7.1f} MiB of device memory; OS threads {n0} -> {len(os.listdir('/proc/self/task'))}", flush=True)
