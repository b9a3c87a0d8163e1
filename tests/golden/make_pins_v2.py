"""Regression pins, second set (round 2; like pins_v1 these are pins of THIS implementation, not reference-derived): Vamana
GreedySearch, the filtered search, the recompute provider (features, weights, embeddings), the feature-space distance of the
recompute-on graph search, the cross-shard merge and the hybrid orchestration — produced by the oracle when the GPU path was
validated against it.  The oracle (CPU suite) and the HIP path (GPU suite) must keep reproducing them bit for bit.
Run: python tests/golden/make_pins_v2.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyoracle as po

SEED = 0x5EED0001


def build():
    out = {}
    X = po.gen_rows(SEED, 96, 32, 64, 1.0, 0, 0, 2000)
    Q = po.gen_rows(SEED, 96, 32, 64, 1.0, 1, 0, 16)
    # Vamana: sequential build (alpha 1.2), GreedySearch at two beam widths
    V = po.Graph.build_vamana(X, R=12, L=32, alpha=1.2, two_stage=False)  # the pins predate the two-stage prune
    lv, uo, a0, aU = V.export()
    out["vam_graph"] = np.array([int(a0.astype(np.uint64).sum()), V.entry, int((a0 != 0xFFFFFFFF).sum())], np.uint64)
    for L in (8, 40):
        k, dd, c, st = V.search_batch(Q, 6, L, 1, 1)
        out[f"vam_keys_L{L}"], out[f"vam_dists_L{L}"], out[f"vam_stats_L{L}"] = k, dd.view(np.uint32), st
    # filtered HNSW search: allow every third position
    G = po.Graph.build_hnsw(X, M=8, efc=32)
    allow = np.packbits((np.arange(2000) % 3) == 0, bitorder="little")
    k, dd, c, st = G.search_filtered_batch(Q, 5, 24, allow, 0, 1)
    out["filt_keys"], out["filt_dists"], out["filt_counts"] = k, dd.view(np.uint32), c
    # recompute provider: synthetic features / weights, embeddings, feature-space projection
    F = po.synth_features(SEED, 64, 16, 1.0, 0, 5, 40)
    W = po.synth_weights(SEED, 64, 96)
    E = po.recompute_encode(F, W)
    out["rc_features"], out["rc_weights_crc"] = F[:4], np.array([int(W.astype(np.uint64).sum())], np.uint64)
    out["rc_embed"] = E[:8].view(np.uint32)
    out["rc_project"] = po.project_queries(W, E[:3], 64).view(np.uint32)
    # cross-shard merge of three lists with a tie in distance (resolved by key) and a short list
    mk = np.array([[5, 9, 40], [7, 8, 41], [1, 2, 3]], np.uint64)
    md = np.array([[0.1, 0.3, 0.5], [0.1, 0.2, 0.9], [0.4, 0.45, 0.0]], np.float32)
    mc = np.array([3, 3, 2], np.uint32)
    rk, rd = po.merge_topk(mk, md, mc, 5)
    out["merge_keys"], out["merge_dists"] = rk, rd.view(np.uint32)
    return out


if __name__ == "__main__":
    out = build()
    np.savez_compressed(os.path.join(HERE, "pins_v2.npz"), **out)
    print({k: v.shape for k, v in out.items()})
