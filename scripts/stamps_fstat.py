"""Diagnostic (scripts/stamps.sh build): where does fused_fstat_kernel spend its cycles?  Prints phase shares."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import leann_rs_amd as la
L, chk = la.lib(), la._native.check
n, h, d, nq, k = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000, 256, 768, 64, 10
dev = torch.device("cuda", 0)
F = torch.empty((n, h), dtype=torch.int16, device=dev); W = torch.empty((h, d), dtype=torch.int16, device=dev)
chk(L.leann_synth_features_device(0x5EED0001, h, 64, 4096, 1.0, 0, 0, n, F.data_ptr(), None))
chk(L.leann_synth_weights_device(0x5EED0001, h, d, W.data_ptr(), None))
Q = torch.empty((nq, d), dtype=torch.float32, device=dev)
chk(L.leann_synth_rows_device(0x5EED0001, d, d, 64, 4096, 1.0, 1, 0, nq, Q.data_ptr(), None)); torch.cuda.synchronize()
r = C.c_void_p(); chk(L.leann_recompute_create(F.data_ptr(), n, h, W.data_ptr(), d, 0, 0, C.byref(r)))
keys = torch.empty((nq, k), dtype=torch.int64, device=dev); sc = torch.empty((nq, k), dtype=torch.float32, device=dev); cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
fn = L.leann_debug_fstat_stamps; fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
for it in range(2):
    fn(None, 1)
    chk(L.leann_recompute_search_batch_device(r, Q.data_ptr(), nq, k, None, keys.data_ptr(), sc.data_ptr(), cnt.data_ptr(), None)); torch.cuda.synchronize()
out = (C.c_uint64 * 16)(); fn(out, 0)
v = np.array(list(out), dtype=np.float64); tot = v[:6].sum()
names = ["wait own DMA/stores/F", "barrier", "issue DMA", "", "compute W", "compute G + stores", "", "", "", "", "", "", "  last W visit incl. norms (its MFMA loop is also counted in compute W)", "  G MFMA loop, plain (part of compute G)", "  G MFMA loop of the last score visit, with feature prefetch", ""]
ms3 = (C.c_float * 3)(); L.leann_recompute_last_timing(r, ms3)
print(f"n={n}: fused {ms3[0]:.2f} ms, topk {ms3[2]:.2f} ms; sub-slice visits (wave 0 of each WG): {int(v[6])}")
for nm, x in zip(names, v):
    if nm: print(f"  {nm:24s} {x / tot * 100:5.1f} %   {x / max(v[6], 1):9.0f} cycles per sub-slice visit")
