"""Where does a Vamana graph of degree R stop being navigable?  (R = 32 reaches recall@10 0.98 at 1M x 1536
and 0.60 at 10M.)

    python scripts/exp/vamana_scale.py --rows 1000000,2000000,4000000,10000000 --d 256 --R 32 --variants base,p2,p2a100,hnsw16

For every size and variant: build on the GPU, recall@10 at several beams against the exact scan, evaluations / hops per query,
out-degree and in-degree statistics of the exported graph.  Variants set the builder's environment knobs for that build:
  base     one pass, 4 pending back-edges (the default)
  strict   LEANN_VAMANA_PENDING=0
  p2       LEANN_VAMANA_PASSES=2
  p2a100   two passes, first with alpha = 1.0 (DiskANN's schedule)
  hnswN    HNSW with M = N (level-0 degree 2N) — the same rows under a hierarchy
  any KEY=VALUE[+KEY=VALUE...] list
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import leann_rs_amd as la  # noqa: E402

L, chk = la.lib(), la._native.check
SEED = 0x5EED0001
VARIANTS = {
    "base": {},
    "strict": {"LEANN_VAMANA_PENDING": "0"},
    "p2": {"LEANN_VAMANA_PASSES": "2"},
    "p2a100": {"LEANN_VAMANA_PASSES": "2", "LEANN_VAMANA_ALPHA1_PCT": "100"},
    "p3": {"LEANN_VAMANA_PASSES": "3"},
}
KNOBS = ("LEANN_VAMANA_PENDING", "LEANN_VAMANA_PASSES", "LEANN_VAMANA_ALPHA1_PCT", "LEANN_BUILD_BATCH_FRACTION", "LEANN_VAMANA_NAV",
         "LEANN_VAMANA_LONG", "LEANN_VAMANA_ALPHA_PCT", "LEANN_VAMANA_RANDOM_INIT", "LEANN_VAMANA_TWO_STAGE")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="1000000,2000000,4000000,10000000")
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--R", type=int, default=32)
    ap.add_argument("--efc", type=int, default=128)
    ap.add_argument("--beams", default="128,256,512")
    ap.add_argument("--nq", type=int, default=2000)
    ap.add_argument("--variants", default="base")
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--degrees", action="store_true", help="export the graph and report degree statistics")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    d, k, nq = a.d, 10, a.nq
    ld = (d + 3) // 4 * 4
    sizes = [int(x) for x in a.rows.split(",")]
    nmax = max(sizes)
    X = la.DeviceArray((nmax, ld), np.float32)
    chk(L.leann_synth_rows_device(SEED, d, ld, 64, 4096, a.sigma, 0, 0, nmax, X.ptr, None))
    Q = la.DeviceArray((nq, ld), np.float32)
    chk(L.leann_synth_rows_device(SEED, d, ld, 64, 4096, a.sigma, 1, 0, nq, Q.ptr, None))
    la.sync()
    ok, od, oc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    st = la.DeviceArray((nq, 4), np.uint32)
    results = []
    for n in sizes:
        chk(L.leann_scan_topk_device(X.ptr, n, d, ld, Q.ptr, nq, k, None, 0, ok.ptr, od.ptr, oc.ptr, None))
        la.sync()
        truth = ok.to_host()
        for v in a.variants.split(","):
            for kn in KNOBS:
                os.environ.pop(kn, None)
            backend, R = 1, a.R
            if v.startswith("hnsw"):
                backend, R = 0, int(v[4:])
            elif v in VARIANTS:
                os.environ.update(VARIANTS[v])
            else:
                os.environ.update(dict(kv.split("=") for kv in v.split("+")))
            t0 = time.time()
            s = la.BackendSearcher.build_device(backend, X.ptr, n, d, ld, R, a.efc)
            la.sync()
            build_s = time.time() - t0
            rec = {"n": n, "d": d, "variant": v, "R": R, "efc": a.efc, "build_s": round(build_s, 1), "beams": {}}
            for beam in [int(x) for x in a.beams.split(",")]:
                s.search_batch_device(Q.ptr, nq, k, beam, ok.ptr, od.ptr, oc.ptr, st.ptr, None)
                la.sync()
                got, stats = ok.to_host(), st.to_host().astype(np.int64)
                r10 = float(np.mean([len(set(got[i].tolist()) & set(truth[i].tolist())) / k for i in range(nq)]))
                r1 = float(np.mean(got[:, 0] == truth[:, 0]))
                rec["beams"][beam] = {"recall10": round(r10, 4), "recall1": round(r1, 4), "evals": float(stats[:, 0].mean()),
                                      "hops": float((stats[:, 1] + stats[:, 2]).mean())}
            if a.degrees:
                g = s.graph_export()
                adj = g["adj0"]
                valid = adj != 0xFFFFFFFF
                outdeg = valid.sum(1)
                indeg = np.bincount(adj[valid].astype(np.int64), minlength=n)
                rec["deg"] = {"out_mean": float(outdeg.mean()), "out_full_frac": float((outdeg == adj.shape[1]).mean()),
                              "in_zero_frac": float((indeg == 0).mean()), "in_le2_frac": float((indeg <= 2).mean()),
                              "in_p50": float(np.percentile(indeg, 50)), "in_p99": float(np.percentile(indeg, 99)), "in_max": int(indeg.max())}
                del g, adj, valid
            s.close()
            print(json.dumps(rec), flush=True)
            results.append(rec)
    if a.out:
        with open(a.out, "w") as f:
            for r in results:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
