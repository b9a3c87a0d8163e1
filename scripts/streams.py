"""Small-batch serving: batch-64 calls issued round-robin on S HIP streams (re-entrant C ABI)."""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import leann_rs_amd as la
L, chk = la.lib(), la._native.check
n, d, ef, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 768, 128, 10
dev = torch.device("cuda", 0)
X = torch.empty((n, d), dtype=torch.float32, device=dev)
chk(L.leann_synth_rows_device(0x5EED0001, d, d, 64, 4096, 1.0, 0, 0, n, X.data_ptr(), None)); torch.cuda.synchronize()
s = la.BackendSearcher.build_device(0, X.data_ptr(), n, d, d, 32, 128)
NQ = 16384
Q = torch.empty((NQ, d), dtype=torch.float32, device=dev)
chk(L.leann_synth_rows_device(0x5EED0001, d, d, 64, 4096, 1.0, 1, 0, NQ, Q.data_ptr(), None)); torch.cuda.synchronize()
keys = torch.empty((NQ, k), dtype=torch.int64, device=dev); dists = torch.empty((NQ, k), dtype=torch.float32, device=dev); cnt = torch.empty((NQ,), dtype=torch.int32, device=dev)
for B in (1, 8, 64, 256, 1024):
    for S in (1, 4, 16, 32):
        streams = [torch.cuda.Stream() for _ in range(S)]
        calls = NQ // B
        if calls > 4096: calls = 4096
        def run():
            for c in range(calls):
                st = streams[c % S]
                s.search_batch_device(Q.data_ptr() + c * B * d * 4, B, k, ef, keys.data_ptr() + c * B * k * 8, dists.data_ptr() + c * B * k * 4, cnt.data_ptr() + c * B * 4, None, C.c_void_p(st.cuda_stream))
        run(); torch.cuda.synchronize()
        t = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"batch {B:5d} x {S:2d} streams: {calls*B/dt:10.0f} QPS  ({dt/calls*1e6:8.1f} us per call issued, {calls} calls)", flush=True)
