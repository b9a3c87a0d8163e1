"""Builds profiles/r01_hnsw10m_profile.md + pmc_traffic_hnsw10m.json from the rocprofv3 outputs under gpurun_out/<prefix>_*."""
import csv, json, statistics, shutil, sys
pfx = sys.argv[1] if len(sys.argv) > 1 else "p4"
root = "gpurun_out/"
stats = list(csv.DictReader(open(root + pfx + "_stats/r01_kernel_stats.csv")))
b = json.load(open(root + pfx + "_stats.json")); bf = json.load(open(root + pfx + "_fetch.json")); bw = json.load(open(root + pfx + "_write.json"))
un = json.load(open(root + "bench_default.json"))
assert b["config"]["ef_search"] == bf["config"]["ef_search"] == bw["config"]["ef_search"] == un["config"]["ef_search"]
trace = list(csv.DictReader(open(root + pfx + "_stats/r01_kernel_trace.csv")))
q = [r for r in trace if "beam_search_kernel<3, 4, 4, false>" in r["Kernel_Name"]]
q.sort(key=lambda r: int(r["Dispatch_Id"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in q[-20:]]
def lastn(f, n):
    rr = list(csv.DictReader(open(f))); rr.sort(key=lambda r: int(r["Dispatch_Id"])); return [float(r["Counter_Value"]) for r in rr[-n:]]
f = lastn(root + pfx + "_fetch/r01_counter_collection.csv", 4); w = lastn(root + pfx + "_write/r01_counter_collection.csv", 4)
fe = sum(f) / 4 * 1024; wr = sum(w) / 4 * 1024; alg = bf["roofline"]["algorithmic_bytes_per_launch"]
ef = b["config"]["ef_search"]
json.dump({"workload": "hnsw10m", "ef_search": ef, "ef_construction": b["config"]["ef_construction"], "kernel": "beam_search_kernel<3,4,4,false>",
           "launch": "16384 queries, k=10, 10M x 768 f32", "FETCH_SIZE_KB_per_launch": sum(f) / 4, "WRITE_SIZE_KB_per_launch": sum(w) / 4,
           "fetch_bytes_raw": fe, "fetch_bytes_corrected_x2_gfx950": 2 * fe, "write_bytes": wr, "hbm_bytes_per_launch": 2 * fe + wr,
           "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (2 * fe + wr) / alg,
           "method": "two separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE), --kernel-include-regex on the query kernel, last 4 launches of "
                     "`bench.py --no-cpu-baseline --steps 4 --warmup 1`; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 reports half the "
                     "bytes of 16 B/lane coalesced reads); units KB*1024"}, open("profiles/pmc_traffic_hnsw10m.json", "w"), indent=1)
shutil.copy(root + pfx + "_stats/r01_kernel_stats.csv", "profiles/r01_hnsw10m_kernel_stats.csv")
shutil.copy(root + pfx + "_stats/r01_domain_stats.csv", "profiles/r01_hnsw10m_domain_stats.csv")
shutil.copy(root + "bench_default.json", "profiles/r01_hnsw10m_bench.json")
qrow = [r for r in stats if "beam_search_kernel<3, 4, 4, false>" in r["Name"]][0]
md = f"""# Round 1 profile — bench.py default workload (hnsw10m, efc={b['config']['ef_construction']}, --ef auto -> ef={ef}), 1x MI355X

Command (GPU box): `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/{pfx}_stats -o r01 -- python3 bench.py --no-cpu-baseline`
Files: `r01_hnsw10m_kernel_stats.csv` (all kernels of the process incl. index construction), `r01_hnsw10m_domain_stats.csv`,
`pmc_traffic_hnsw10m.json` (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes), `r01_hnsw10m_bench.json` (un-profiled bench line).

## Dominant kernel of the timed region: `beam_search_kernel<3, 4, 4, false>`
(T=3 chunks of 256 floats, R=4 rows in flight per wave, 4 waves per query; `<…, true>` / `<3,4,16,true>` are the index-construction instantiations)

| quantity | value |
|---|---|
| kernel-stats row (`r01_hnsw10m_kernel_stats.csv`) | Calls {qrow['Calls']} (ef ladder + recall + 3 warm-up + 20 timed launches of 16 384 queries), AverageNs {float(qrow['AverageNs']):.0f} |
| rocprofv3 kernel-trace average of the 20 timed dispatches (same run) | **{statistics.mean(d):.3f} ms** (min {min(d):.3f}, max {max(d):.3f}) |
| bench.py HIP-event average (same profiled run) | **{b['roofline']['kernel_avg_ms']:.3f} ms** |
| bench.py HIP-event average, un-profiled (`r01_hnsw10m_bench.json`) | {un['roofline']['kernel_avg_ms']:.3f} ms -> {un['value']:.0f} queries/s, recall@10 {un['recall_at_10']:.4f} |
| launch geometry | 16 384 workgroups x 256 threads; compile-time 99 VGPR, 35 KB dynamic LDS, no scratch -> 4 workgroups per CU (LDS) |
| algorithmic bytes per launch (n_evals*768*4 + hops0*64*4 + hopsU*32*4, counted by the kernel) | {un['roofline']['algorithmic_bytes_per_launch']/1e9:.2f} GB ({un['roofline']['dist_evals_per_query']:.0f} distance evaluations + {un['roofline']['hops_per_query']:.0f} hops per query) |
| achieved | {un['roofline']['achieved']:.0f} GB/s = **{un['roofline']['frac']*100:.1f} % of 8000 GB/s** |
| PMC FETCH_SIZE per launch (raw / x2 gfx950 correction) | {fe/1e9:.2f} GB / {2*fe/1e9:.2f} GB |
| PMC WRITE_SIZE per launch | {wr/1e6:.1f} MB |
| HBM traffic / algorithmic bytes | **{(2*fe+wr)/alg:.3f}** (no wasted re-reads; the hot upper-level rows hit L2 / Infinity Cache) |

(The `Calls`/`AverageNs` row also averages the shorter ef-ladder launches; the trace isolates the timed ones. `r01_kernel_trace.csv` is 8 MB and not committed.)

Other operating points of the same kernel (un-profiled bench lines, graphs built before the permuted insertion order): efc=128 graph, ef=80: 905 k queries/s, recall 0.960, 75.4 %;
efc=128, ef=128 (BASELINE configs[1] beam): 702 k queries/s, recall 0.979, 71.5 %, PMC traffic 0.993x algorithmic.

## In-kernel stamps (diagnostic build `scripts/stamps.sh`, shares only): cycles per hop seen by wave 0, ef=64, 4M rows
| phase | batch 64 (idle chip) | what |
|---|---|---|
| B | 2 270 (26 %) | adjacency list load + visited-table filter |
| C | 3 250 (36 %) | row gather (one HBM round trip) + wave-tree reductions (DPP ladder; 3 800 with ds_bpermute shuffles) |
| D | 1 620 (18 %) | merge by rank (was 7 700 before the 16-B pipelined scan, 2 100 before the DPP wave minimum) |
| barriers | 1 770 (20 %) | three workgroup barriers |

## Index construction (not in the timed region; {un['config']['index_build_s']:.0f} s for 10M x 768 at efc={b['config']['ef_construction']})
`beam_search_kernel<3,4,4,true>` (construction searches) dominates; `select_kernel` and `reverse_merge_kernel` ~20 % each.
"""
open("profiles/r01_hnsw10m_profile.md", "w").write(md)
print(md[:2400])
