"""The oracle's SEQUENTIAL Vamana builder (oracle/oracle.c:orc_vamana_build — random R-regular start graph, two passes over the points in
random order, alpha 1.0 then 1.2, RobustPrune over the visited set: DiskANN Alg. 1-3 point by point) on the same synthetic rows as
scripts/exp/vamana_scale.py, CPU only.  Answers VERDICT r2 item 2's alternative: does a sequential builder show the same fall of
recall@10 with the corpus size at R = 32 as the GPU's batched builder?

    python scripts/exp/vamana_seq_oracle.py --rows 1000000 --d 256 --R 32 --L 128 --out profiles/r03_vamana_seq_oracle_1m.json

Single-threaded by nature (every insertion sees the previous one): ~1 h per million rows at d = 256.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--R", type=int, default=32)
    ap.add_argument("--L", type=int, default=128)
    ap.add_argument("--nq", type=int, default=2000)
    ap.add_argument("--beams", default="128,256,512")
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--fast-dot", action="store_true", help="plain AVX2 dot instead of the canonical tree (not bit-compatible with the GPU; same graph quality)")
    ap.add_argument("--paper-prune", action="store_true", help="one-stage RobustPrune of the paper (Alg. 2) instead of DiskANN's two-stage occlude_list")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    n, d, k = a.rows, a.d, 10
    t0 = time.time()
    X = np.concatenate([po.gen_rows(0x5EED0001, d, 64, 4096, 1.0, 0, i0, min(250_000, n - i0)) for i0 in range(0, n, 250_000)])
    Q = po.gen_rows(0x5EED0001, d, 64, 4096, 1.0, 1, 0, a.nq)
    print(f"rows generated in {time.time() - t0:.0f}s", flush=True)
    # exact top-10 by BLAS (f32 GEMM; near-ties do not move a recall figure)
    best = np.full((a.nq, k), -np.inf, np.float32)
    besti = np.zeros((a.nq, k), np.int64)
    for r0 in range(0, n, 100_000):  # 2000 x 100k scores + the index array of argpartition: ~2.5 GB at a time
        S = Q @ X[r0:r0 + 100_000].T
        idx = np.argpartition(-S, k, axis=1)[:, :k]
        sc = np.take_along_axis(S, idx, 1)
        allsc, allid = np.concatenate([best, sc], 1), np.concatenate([besti, idx + r0], 1)
        o = np.argsort(-allsc, axis=1, kind="stable")[:, :k]
        best, besti = np.take_along_axis(allsc, o, 1), np.take_along_axis(allid, o, 1)
    truth = besti
    if a.fast_dot:
        po.lib().orc_set_fast_dot(1)
    t0 = time.time()
    G = po.Graph.build_vamana(X, R=a.R, L=a.L, alpha=1.2, two_stage=not a.paper_prune)
    build_s = time.time() - t0
    print(f"sequential build of {n} x {d}, R={a.R}, L={a.L}: {build_s:.0f}s", flush=True)
    rec = {"builder": "oracle/oracle.c:orc_vamana_build (sequential, random start graph, 2 passes: alpha 1.0 then 1.2), RobustPrune "
                      + ("one-stage (paper Alg. 2)" if a.paper_prune else "two-stage (DiskANN occlude_list)"), "n": n, "d": d, "R": a.R,
           "L_build": a.L, "build_s": round(build_s), "beams": {}}
    for beam in [int(x) for x in a.beams.split(",")]:
        kk, dd, cc, st = G.search_batch(Q, k, beam, 1, a.threads)
        r10 = float(np.mean([len(set(kk[i].tolist()) & set(truth[i].tolist())) / k for i in range(a.nq)]))
        r1 = float(np.mean(kk[:, 0].astype(np.int64) == truth[:, 0]))
        rec["beams"][beam] = {"recall10": round(r10, 4), "recall1": round(r1, 4), "evals": float(st[:, 0].mean()), "hops": float(st[:, 1].mean())}
    adj = G.export()[2]
    valid = adj != 0xFFFFFFFF
    indeg = np.bincount(adj[valid].astype(np.int64), minlength=n)
    rec["deg"] = {"out_mean": float(valid.sum(1).mean()), "in_zero_frac": float((indeg == 0).mean()), "in_p50": float(np.percentile(indeg, 50)),
                  "in_p99": float(np.percentile(indeg, 99)), "in_max": int(indeg.max())}
    print(json.dumps(rec), flush=True)
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
