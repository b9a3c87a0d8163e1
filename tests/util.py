import numpy as np

# synthetic-set defaults: seed, r, clusters, sigma
SEED = 0x5EED0001
GEN = dict(r=64, n_clusters=4096, sigma=1.0)


def synth(po, n, d, stream=0, i0=0, r=64, n_clusters=256, sigma=1.0, seed=SEED):
    return po.gen_rows(seed, d, r, n_clusters, sigma, stream, i0, n)


def recall_at_k(found, truth):
    k = truth.shape[1]
    hits = 0
    for f, t in zip(found, truth):
        hits += len(set(int(x) for x in f[:k]) & set(int(x) for x in t))
    return hits / (len(truth) * k)


def traced_graph(name="traced_graph_64.json"):
    """tests/golden/traced_graph_64.json (hand-built 64-node graph) or traced_graph_400.json (400 nodes, three levels, duplicated
    vectors) as arrays + the independently traced expectations"""
    import json, os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)))
    X = np.array(fx["vectors"], np.float32)
    levels = np.array(fx["levels"], np.uint8)
    upper_off = np.concatenate([[0], np.cumsum(levels)[:-1]]).astype(np.uint32)
    adj0 = np.array(fx["adj0"], np.uint32)
    adjU = np.array(fx["adjU"], np.uint32)
    return fx, X, levels, upper_off, adj0, adjU


def write_gx1(path, kind, X, M, M0, max_level, entry, levels, upper_off, adj0, adjU, efc=64, alpha=1.2):
    """a version-1 LEANNGX1 index file (csrc/indexfile.hip) written from numpy arrays; returns the header bytes"""
    import struct
    X = np.ascontiguousarray(X, np.float32)
    n, d = X.shape
    adjU = np.ascontiguousarray(adjU, np.uint32).reshape(-1, M) if np.size(adjU) else np.zeros((0, M), np.uint32)
    hd = struct.pack("<8sIIQIIIIIIfIQI60x", b"LEANNGX1", 1, kind, n, d, M, M0, max_level, entry, efc, alpha, 0, adjU.shape[0], 0)
    assert len(hd) == 128
    with open(path, "wb") as f:
        f.write(hd)
        f.write(np.ascontiguousarray(levels, np.uint8).tobytes())
        f.write(np.ascontiguousarray(upper_off, np.uint32).tobytes())
        f.write(np.ascontiguousarray(adj0, np.uint32).tobytes())
        f.write(adjU.tobytes())
        f.write(X.tobytes())
    return hd
