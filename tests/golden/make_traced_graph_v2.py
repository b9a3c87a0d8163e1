"""Second traced fixture (round 2): a 400-node, THREE-level graph over small-integer vectors — built here by exact integer nearest
neighbours, with duplicated vectors (exact distance ties) and a few random long links — searched by the same independent pure-Python
implementation of the textbook algorithms as traced_graph_64 (Malkov & Yashunin Alg. 5 / Alg. 2 with two heaps on (dist, id); DiskANN
Alg. 1 GreedySearch with one sorted list).  No oracle, no numpy arithmetic: every dot product is a small integer, exact in f32 in any
summation order, so the expected ids, distances, expansion orders and evaluation counts do not depend on floating point.
The oracle's two formulations (CPU suite) and the HIP kernel (GPU suite) must reproduce every case.
Run: python tests/golden/make_traced_graph_v2.py"""
import heapq
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
N, D, M, M0 = 400, 24, 6, 12
EMPTY = 0xFFFFFFFF
rng = random.Random(0x5EED0006)

X = [[rng.randint(-3, 3) for _ in range(D)] for _ in range(N)]
for i in range(0, 60, 3):  # exact duplicates: distance ties that only the id can break
    X[N - 1 - i] = list(X[i])
levels = [0] * N
for i in range(N):
    u = rng.random()
    levels[i] = 2 if u < 1 / 36 else (1 if u < 1 / 6 else 0)
levels[7] = 2  # the entry point reaches the top level
entry = 7


def dot(a, b):
    return sum(x * y for x, y in zip(a, b))


def dist(q, i):
    return 1 - dot(q, X[i])


def nearest(i, pool, m):
    cand = sorted((1 - dot(X[i], X[j]), j) for j in pool if j != i)
    return [j for _, j in cand[:m]]


adj = {}  # (node, level) -> neighbour list
for lvl in (2, 1, 0):
    pool = [i for i in range(N) if levels[i] >= lvl]
    cap = M0 if lvl == 0 else M
    for i in pool:
        nb = nearest(i, pool, cap - 2 if len(pool) > cap else cap)
        extra = [j for j in rng.sample(pool, min(len(pool), 4)) if j != i and j not in nb]
        adj[(i, lvl)] = (nb + extra)[:cap]


def search_layer(q, eps, ef, nbrs, trace):
    visited = {e for _, e in eps}
    cand = list(eps)
    heapq.heapify(cand)
    res = [(-d, -e) for d, e in eps]
    heapq.heapify(res)
    while len(res) > ef:
        heapq.heappop(res)
    while cand:
        d, c = heapq.heappop(cand)
        worst = (-res[0][0], -res[0][1])
        if len(res) == ef and (d, c) > worst:
            break
        trace["expanded"].append(c)
        for e in nbrs(c):
            if e in visited:
                continue
            visited.add(e)
            de = dist(q, e)
            trace["evaluated"].append(e)
            worst = (-res[0][0], -res[0][1])
            if len(res) < ef or (de, e) < worst:
                heapq.heappush(cand, (de, e))
                heapq.heappush(res, (-de, -e))
                if len(res) > ef:
                    heapq.heappop(res)
    return sorted((-d, -e) for d, e in res)


def knn(q, k, ef):
    best = (dist(q, entry), entry)
    n_evals, hops_upper = 1, 0
    for lvl in (2, 1):
        t = {"expanded": [], "evaluated": []}
        best = search_layer(q, [best], 1, lambda c, lvl=lvl: adj[(c, lvl)], t)[0]
        n_evals += len(t["evaluated"])
        hops_upper += len(t["expanded"])
    t0 = {"expanded": [], "evaluated": []}
    res = search_layer(q, [best], max(ef, k), lambda c: adj[(c, 0)], t0)
    return {"k": k, "ef": ef, "query": q, "ids": [e for _, e in res[:k]], "dists": [d for d, _ in res[:k]],
            "hops_upper": hops_upper, "expanded_base": t0["expanded"], "n_evals": n_evals + len(t0["evaluated"])}


def vamana(q, k, L):
    L = max(L, k)
    lst = [(dist(q, entry), entry)]
    seen, done, expanded, n_evals = {entry}, set(), [], 1
    while True:
        pending = [x for x in lst if x[1] not in done]
        if not pending:
            break
        d, c = min(pending)
        done.add(c)
        expanded.append(c)
        for e in adj[(c, 0)]:
            if e in seen:
                continue
            seen.add(e)
            n_evals += 1
            lst.append((dist(q, e), e))
        lst = sorted(lst)[:L]
    return {"k": k, "L": L, "query": q, "ids": [e for _, e in lst[:k]], "dists": [d for d, _ in lst[:k]], "expanded": expanded,
            "n_evals": n_evals}


queries = [[rng.randint(-2, 2) for _ in range(D)] for _ in range(10)] + [list(X[0]), list(X[33]), list(X[N - 1])]
cases = [knn(q, k, ef) for q in queries for k, ef in ((10, 24), (1, 1), (5, 40), (12, 7))]
upper_nodes = [i for i in range(N) if levels[i] > 0]
fix = {"n": N, "d": D, "M": M, "M0": M0, "entry": entry, "max_level": 2, "vectors": X, "levels": levels,
       "adj0": [adj[(i, 0)] + [EMPTY] * (M0 - len(adj[(i, 0)])) for i in range(N)],
       # upper lists in node order, level 1 first then level 2 for each node (the flat layout of DESIGN.md §2)
       "adjU": [adj[(i, l)] + [EMPTY] * (M - len(adj[(i, l)])) for i in range(N) for l in range(1, levels[i] + 1)],
       "cases": cases, "vamana_cases": [vamana(q, k, L) for q in queries for k, L in ((10, 24), (3, 2), (8, 48))]}
json.dump(fix, open(os.path.join(HERE, "traced_graph_400.json"), "w"))
print(len(cases), "cases;", sum(levels[i] == 2 for i in range(N)), "nodes on level 2,", sum(levels[i] >= 1 for i in range(N)), "on level 1;",
      cases[0]["ids"], cases[0]["n_evals"], os.path.getsize(os.path.join(HERE, "traced_graph_400.json")), "bytes")
