import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import leann_rs_amd as la, pyoracle as po
SEED = 0x5EED0001
for d, r in ((768, 64), (128, 32), (1536, 64), (768, 64)):
    n = 300; ld = d
    ref = po.gen_rows(SEED, d, r, 97, 0.7, 0, 0, n)
    for it in range(12):
        buf = la.DeviceArray((n, ld), np.float32)
        la._native.check(la.lib().leann_synth_rows_device(SEED, d, ld, r, 97, 0.7, 0, 0, n, buf.ptr, None))
        la.sync()
        got = buf.to_host()
        bad = got.view(np.uint32) != ref.view(np.uint32)
        print(d, r, it, "mismatch elems", int(bad.sum()), "rows", int(bad.any(1).sum()), "first rows", np.nonzero(bad.any(1))[0][:8].tolist(),
              "cols of row", (np.nonzero(bad[np.nonzero(bad.any(1))[0][0]])[0][:6].tolist() if bad.any() else []),
              "maxabs", float(np.abs(got-ref).max()))
