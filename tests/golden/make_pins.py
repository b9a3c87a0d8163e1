"""Regression pins of THIS implementation (not reference-derived): bits of the synthetic generator, canonical dot
products, levels, and one small HNSW search, produced by the oracle at the time the GPU path was first validated
against it.  Both the oracle (CPU suite) and the HIP path (GPU suite) must keep reproducing them bit for bit.
Run: python tests/golden/make_pins.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyoracle as po

SEED = 0x5EED0001
out = {}
for name, (d, r, C_, sig, stream, i0) in {"c768": (768, 64, 4096, 1.0, 0, 0), "q768": (768, 64, 4096, 1.0, 1, 7),
                                           "iid128": (128, 0, 1, 0.0, 0, 3), "c1536": (1536, 64, 4096, 1.0, 0, 10 ** 7)}.items():
    out["gen_" + name] = po.gen_rows(SEED, d, r, C_, sig, stream, i0, 3).view(np.uint32)[:, :16]
X = po.gen_rows(SEED, 96, 32, 64, 1.0, 0, 0, 2000)
Q = po.gen_rows(SEED, 96, 32, 64, 1.0, 1, 0, 16)
out["dot_canon"] = np.array([po.dot(Q[i], X[i], "canon") for i in range(16)], np.float32).view(np.uint32)
out["levels_M32"] = np.array([po.lib().orc_level(0x5EED0003, i, 32) for i in range(4096)], np.uint8)
G = po.Graph.build_hnsw(X, M=8, efc=32)
k, dd, c, st = G.search_batch(Q, 5, 24, 0, 1)
out["hnsw_keys"], out["hnsw_dists"], out["hnsw_stats"] = k, dd.view(np.uint32), st
lv, uo, a0, aU = G.export()
out["hnsw_adj0_crc"] = np.array([int(a0.astype(np.uint64).sum()), int(aU.astype(np.uint64).sum()), G.entry, G.max_level], np.uint64)
np.savez_compressed(os.path.join(HERE, "pins_v1.npz"), **out)
print({k: v.shape for k, v in out.items()})
