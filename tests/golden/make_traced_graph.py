"""Golden fixture (4) of SURVEY.md §8c: a 64-node hand-built two-level HNSW graph over small-integer vectors with the expected
visit order of the textbook search, traced by an independent pure-Python implementation (no oracle, no numpy arithmetic: all
dot products are small integers, exact in f32 whatever the summation order, so the trace does not depend on floating point).

Graph: nodes 0..63 on an 8 x 8 grid; vector of node (r, c) = one-hot(row r) * 3 + one-hot(column c) * 2 in 16 dimensions
(integer coordinates); level-0 neighbours = the 4 grid neighbours + the node mirrored through the centre (cap 8); nodes whose
row and column are both multiples of 4 also live on level 1, fully connected among themselves (cap 4); entry = node 0.
Search = Malkov & Yashunin Alg. 5 / Alg. 2: greedy (ef = 1) on level 1, ef-beam on level 0, metric 1 - <q, x>, every
comparison on (dist, id).  Run: python tests/golden/make_traced_graph.py"""
import heapq
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
N, D, M, M0 = 64, 16, 4, 8
EMPTY = 0xFFFFFFFF


def vec(i):
    r, c = divmod(i, 8)
    v = [0] * D
    v[r] += 3
    v[8 + c] += 2
    return v


X = [vec(i) for i in range(N)]
adj0 = []
for i in range(N):
    r, c = divmod(i, 8)
    nb = []
    for dr, dc in ((-1, 0), (1, 0), (0, -1), (0, 1)):
        rr, cc = r + dr, c + dc
        if 0 <= rr < 8 and 0 <= cc < 8:
            nb.append(rr * 8 + cc)
    mirror = (7 - r) * 8 + (7 - c)
    if mirror != i and mirror not in nb:
        nb.append(mirror)
    adj0.append(nb)
upper = [i for i in range(N) if (i // 8) % 4 == 0 and (i % 8) % 4 == 0]  # 0, 4, 32, 36
levels = [1 if i in upper else 0 for i in range(N)]
adjU = {i: [j for j in upper if j != i] for i in upper}


def dist(q, i):
    return 1 - sum(a * b for a, b in zip(q, X[i]))


def search_layer(q, eps, ef, nbrs, trace):
    """Alg. 2 with a candidate min-heap and a result max-heap on (dist, id)."""
    visited = {e for _, e in eps}
    cand = list(eps)
    heapq.heapify(cand)
    res = [(-d, -e) for d, e in eps]  # max-heap on (dist, id) via negation
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
    trace = {"expanded": [], "evaluated": [0]}
    best = (dist(q, 0), 0)
    t1 = {"expanded": [], "evaluated": []}
    best = search_layer(q, [best], 1, lambda c: adjU[c], t1)[0]
    t0 = {"expanded": [], "evaluated": []}
    res = search_layer(q, [best], max(ef, k), lambda c: adj0[c], t0)
    return {"k": k, "ef": ef, "query": q, "ids": [e for _, e in res[:k]], "dists": [d for d, _ in res[:k]],
            "expanded_upper": t1["expanded"], "expanded_base": t0["expanded"],
            "n_evals": 1 + len(t1["evaluated"]) + len(t0["evaluated"])}


def greedy_search(q, start, L):
    """DiskANN / Vamana Alg. 1 (GreedySearch) on the level-0 graph: one list of at most L entries sorted by (dist, id); repeatedly
    expand the closest entry that has not been expanded yet."""
    lst = [(dist(q, start), start)]
    seen, done, expanded, n_evals = {start}, set(), [], 1
    while True:
        pending = [x for x in lst if x[1] not in done]
        if not pending:
            break
        d, c = min(pending)
        done.add(c)
        expanded.append(c)
        for e in adj0[c]:
            if e in seen:
                continue
            seen.add(e)
            n_evals += 1
            lst.append((dist(q, e), e))
        lst = sorted(lst)[:L]
    return lst, expanded, n_evals


def vamana(q, k, L):
    L = max(L, k)
    lst, expanded, n_evals = greedy_search(q, 0, L)
    return {"k": k, "L": L, "query": q, "ids": [e for _, e in lst[:k]], "dists": [d for d, _ in lst[:k]], "expanded": expanded,
            "n_evals": n_evals}


queries = [vec(63), vec(27), [1 if j in (2, 13) else 0 for j in range(D)], [2, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 3, 0, 0, 0, 0]]
cases = [knn(q, k, ef) for q in queries for k, ef in ((5, 8), (3, 3), (10, 16))]
fix = {"n": N, "d": D, "M": M, "M0": M0, "entry": 0, "max_level": 1, "vectors": X, "levels": levels,
       "adj0": [nb + [EMPTY] * (M0 - len(nb)) for nb in adj0],
       "upper_nodes": upper, "adjU": [adjU[i] + [EMPTY] * (M - len(adjU[i])) for i in upper], "cases": cases,
       # the level-0 graph alone, searched as a Vamana index from node 0 (DiskAnnSearcher: beam = max(complexity, k), diskann.rs:54)
       "vamana_cases": [vamana(q, k, L) for q in queries for k, L in ((5, 8), (3, 2), (10, 16))]}
json.dump(fix, open(os.path.join(HERE, "traced_graph_64.json"), "w"))
print(len(cases), "cases;", cases[0]["expanded_base"][:10], cases[0]["ids"])
