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
def cycle_sharded(what):
    """round 3: composite handles with registered filters / exact filtered search / save, the sharded recompute search, the hybrid rerank"""
    import ctypes as C, tempfile, shutil
    n, G = X.shape[0], 4
    lows = [((n * g) // G) & ~63 for g in range(G)] + [n]
    s = la.ShardedIndex.build_device(0, [dX.ptr + lows[g] * 128 * 4 for g in range(G)], [lows[g + 1] - lows[g] for g in range(G)], 128, 128, 16, 64,
                                     [0] * G).as_backend()
    allow = np.packbits(rng.random(n) < 0.02, bitorder="little")
    if "filter" in what:
        f = s.register_filter(allow)
        s.search_filter_batch(X[:40], 10, 64, f, mode="auto"); s.search_filter_batch(X[:40], 10, 64, f, mode="walk")
        s.search_filtered_exact_batch(X[:40], 10, allow)
        f.close()
    if "save" in what:
        d = tempfile.mkdtemp()
        s.save(os.path.join(d, "documents.leann"))
        s2 = la.BackendSearcher.load(0, os.path.join(d, "documents.leann"), 128, device="0,0,0,0"); s2.search_batch(X[:8], 5, 32); s2.close()
        shutil.rmtree(d)
    if "recompute" in what:
        L, chk = la.lib(), la._native.check
        F = la.DeviceArray.from_host((rng.integers(0x3000, 0x4000, size=(20000, 256))).astype(np.uint16)); W = la.DeviceArray.from_host(rng.integers(0x3000, 0x4000, size=(256, 128)).astype(np.uint16))
        parts = []
        for g in range(2):
            r = C.c_void_p(); chk(L.leann_recompute_create(F.ptr + g * 10000 * 512, 10000, 256, W.ptr, 128, 0, g * 10000, C.byref(r))); parts.append(r)
        comp = C.c_void_p(); chk(L.leann_recompute_create_sharded((C.c_void_p * 2)(*parts), 2, C.byref(comp)))
        k = la.DeviceArray((64, 10), np.uint64); d_ = la.DeviceArray((64, 10), np.float32); c = la.DeviceArray(64, np.uint32)
        chk(L.leann_recompute_search_batch_device(comp, dX.ptr, 64, 10, None, k.ptr, d_.ptr, c.ptr, None)); la.sync()
        L.leann_recompute_close(comp); [L.leann_recompute_close(r) for r in parts]
    s.close()
for what in ("sharded filter", "sharded save", "sharded recompute"):
    cycle_sharded(what); torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(); n0 = len(os.listdir("/proc/self/task"))
    for _ in range(10): cycle_sharded(what)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    print(f"{what:22s}: 10 more cycles cost {(free0 - free1) / 2**20:7.1f} MiB of device memory; OS threads {n0} -> {len(os.listdir('/proc/self/task'))}", flush=True)
for what in ("build only", "device", "staged", "zero", "threads", "staged zero threads"):
    cycle(what); torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(); n0 = len(os.listdir("/proc/self/task"))
    for _ in range(20): cycle(what)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    print(f"{what:22s}: 20 more cycles cost {(free0 - free1) / 2**20:7.1f} MiB of device memory; OS threads {n0} -> {len(os.listdir('/proc/self/task'))}", flush=True)
