"""Turns the outputs of scripts/regen_profiles.sh, profile_headline.sh and profile_side.sh (gpurun_out/<ROUND>/) into the committed
profiles/<ROUND>_* files: bench lines, rocprofv3 kernel-stats CSVs, the PMC traffic JSON bench.py quotes, and <ROUND>_headline_profile.md."""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r03")  # outputs of the round's runs live under gpurun_out/<ROUND>/, committed copies are profiles/<ROUND>_*
src, dst = os.path.join(ROOT, "gpurun_out", ROUND), os.path.join(ROOT, "profiles")


def last_n(path, name, n):
    rr = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == name]
    rr.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rr[-n:]]


def main():
    for f in sorted(glob.glob(os.path.join(src, "*_bench.json"))):
        shutil.copy(f, os.path.join(dst, ROUND + "_" + os.path.basename(f)))
    for a, b in (("r2_gather_bw.txt", "r02_gather_ceiling.txt"), ("r2_stamps_feat.txt", "r02_stamps_feat.txt")):
        p = os.path.join(ROOT, "gpurun_out", a)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, b))
    for wl in ("recompute10m_graph", "recompute10m"):
        p = os.path.join(src, f"prof_{wl}", ROUND + "_kernel_stats.csv")
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, f"{ROUND}_{wl}_kernel_stats.csv"))
    st_path = os.path.join(src, "prof_stats", ROUND + "_kernel_stats.csv")
    if not os.path.exists(st_path):
        print(f"no headline profile under gpurun_out/{ROUND}/prof_stats")
        return
    shutil.copy(st_path, os.path.join(dst, ROUND + "_hnsw10m_kernel_stats.csv"))
    stats = list(csv.DictReader(open(st_path)))
    trace = [r for r in csv.DictReader(open(os.path.join(src, "prof_stats", ROUND + "_kernel_trace.csv"))) if "beam_search_kernel<3, 4, 4, false>" in r["Kernel_Name"]]
    trace.sort(key=lambda r: int(r["Dispatch_Id"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in trace[-20:]]
    b = json.load(open(os.path.join(src, "prof_stats.json")))
    bf = json.load(open(os.path.join(src, "prof_FETCH_SIZE.json")))
    fe = sum(last_n(os.path.join(src, "prof_FETCH_SIZE", ROUND + "_counter_collection.csv"), "FETCH_SIZE", 4)) / 4 * 1024
    wr = sum(last_n(os.path.join(src, "prof_WRITE_SIZE", ROUND + "_counter_collection.csv"), "WRITE_SIZE", 4)) / 4 * 1024
    hit = sum(last_n(os.path.join(src, "prof_TCC_HIT_sum_TCC_MISS_sum", ROUND + "_counter_collection.csv"), "TCC_HIT_sum", 4)) / 4
    miss = sum(last_n(os.path.join(src, "prof_TCC_HIT_sum_TCC_MISS_sum", ROUND + "_counter_collection.csv"), "TCC_MISS_sum", 4)) / 4
    alg = bf["roofline"]["algorithmic_bytes_per_launch"]
    ef = b["config"]["ef_search"]
    assert ef == bf["config"]["ef_search"]
    json.dump({"workload": "hnsw10m", "ef_search": ef, "ef_construction": b["config"]["ef_construction"], "kernel": "beam_search_kernel<3,4,4,false>",
               "launch": "16384 queries, k=10, 10M x 768 f32", "round": ROUND, "FETCH_SIZE_KB_per_launch": fe / 1024, "WRITE_SIZE_KB_per_launch": wr / 1024,
               "fetch_bytes_raw": fe, "fetch_bytes_corrected_x2_gfx950": 2 * fe, "write_bytes": wr, "hbm_bytes_per_launch": 2 * fe + wr,
               "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (2 * fe + wr) / alg, "TCC_HIT_sum": hit, "TCC_MISS_sum": miss,
               "l2_hit_rate": hit / (hit + miss),
               "method": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; TCC_HIT_sum TCC_MISS_sum), --kernel-include-regex on the query kernel, last 4 "
                         "launches of `bench.py --no-cpu-baseline --no-latency --steps 4 --warmup 1` (scripts/profile_headline.sh); FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md §HBM (gfx950 tallies 128-B requests at 64 B); units KB*1024; the counters sit at the L2<->fabric boundary "
                         "(Infinity-Cache hits included)"},
              open(os.path.join(dst, "pmc_traffic_hnsw10m.json"), "w"), indent=1)
    q = [r for r in stats if "beam_search_kernel<3, 4, 4, false>" in r["Name"]][0]
    rd = ""
    p = os.path.join(src, "prof_rdreq", ROUND + "_counter_collection.csv")
    if os.path.exists(p):
        try:
            a1 = sum(last_n(p, "TCC_EA0_RDREQ_sum", 4)) / 4
            a2 = sum(last_n(p, "TCC_EA0_RDREQ_DRAM_sum", 4)) / 4
            rd = (f"| `TCC_EA0_RDREQ_sum` / `TCC_EA0_RDREQ_DRAM_sum` per launch | {a1:.3e} / {a2:.3e} ({a2 / a1 * 100:.1f} % of the L2's read requests are addressed to "
                  "the DRAM path — which includes the Infinity Cache in front of it; no counter behind it is exposed) |\n")
        except Exception as e:  # noqa: BLE001
            rd = f"| TCC_EA0_RDREQ pass | failed: {e} |\n"
    cm = open(os.path.join(src, "counters_mem.txt")).read() if os.path.exists(os.path.join(src, "counters_mem.txt")) else ""
    import re
    mall = "none (no counter name contains MALL; the TCC_EA0_* family stops at the L2's memory-side interface)" if not re.search(r"Counter_Name\s*:\s*\S*MALL", cm, re.I) else "present: see gpurun_out/{ROUND}/counters_mem.txt"
    md = f"""# {ROUND} profile — bench.py default workload (hnsw10m, efc={b['config']['ef_construction']}, --ef auto -> ef={ef}), 1x MI355X

Commands (GPU box, `scripts/profile_headline.sh`): `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d … -o {ROUND} -- python3 bench.py --headline-only --no-cpu-baseline --no-latency`,
then separate `--pmc` passes (`FETCH_SIZE`; `WRITE_SIZE`; `TCC_HIT_sum TCC_MISS_sum`) of `bench.py --no-cpu-baseline --no-latency --steps 4 --warmup 1`
with `--kernel-include-regex beam_search_kernel`.
Files: `{ROUND}_hnsw10m_kernel_stats.csv` (every kernel of the process incl. index construction), `pmc_traffic_hnsw10m.json`, `{ROUND}_hnsw10m_bench.json` (un-profiled line).

## Dominant kernel of the timed region: `beam_search_kernel<3, 4, 4, false>`

| quantity | value |
|---|---|
| kernel-stats row | Calls {q['Calls']} (ef ladder + recall + warm-up + 20 timed launches of 16 384 queries), AverageNs {float(q['AverageNs']):.0f} |
| rocprofv3 kernel-trace average of the 20 timed dispatches (same run) | **{sum(d) / len(d):.3f} ms** (min {min(d):.3f}, max {max(d):.3f}) |
| bench.py HIP-event average (same profiled run) | **{b['roofline']['kernel_avg_ms']:.3f} ms** -> {b['value']:.0f} queries/s, recall@10 {b['recall_at_10']:.4f} |
| algorithmic bytes per launch (n_evals*768*4 + hops0*64*4 + hopsU*32*4, counted by the kernel) | {alg / 1e9:.2f} GB ({b['roofline']['dist_evals_per_query']:.0f} distance evaluations + {b['roofline']['hops_per_query']:.0f} hops per query) |
| achieved | {b['roofline']['achieved']:.0f} GB/s = **{b['roofline']['frac'] * 100:.1f} % of 8000 GB/s** |
| PMC FETCH_SIZE per launch (raw / x2 gfx950 correction) | {fe / 1e9:.2f} GB / {2 * fe / 1e9:.2f} GB |
| PMC WRITE_SIZE per launch | {wr / 1e6:.1f} MB |
| traffic at the L2<->fabric boundary / algorithmic bytes | **{(2 * fe + wr) / alg:.3f}** (no wasted re-reads) |
| L2: TCC_HIT_sum / TCC_MISS_sum per launch | {hit:.3e} / {miss:.3e} -> hit rate {hit / (hit + miss) * 100:.1f} % ({miss * 128 / 1e9:.1f} GB of 128-B misses = the FETCH_SIZE figure) |
{rd}| Infinity-Cache (MALL) hit counter in `rocprofv3 -L` on this pool | {mall}: the HBM / Infinity-Cache split of those bytes cannot be measured here |
| pure gather ceiling for this row size (`scripts/micro/gather_bw.hip`, 24 GB table, no reuse; `r02_gather_ceiling.txt`) | 6.33-6.43 TB/s for 3 072-B rows: the kernel's {b['roofline']['achieved'] / 1000:.2f} TB/s is the memory system's limit for whole-row gathers |

## Index construction in the same process (not in the timed region; {b['config']['index_build_s']:.1f} s for 10M x 768 at efc={b['config']['ef_construction']})
"""
    for r in stats[:6]:
        md += f"* `{r['Name'][:60]}`: {r['Calls']} calls, {float(r['TotalDurationNs']) / 1e9:.2f} s ({r['Percentage']} %)\n"
    open(os.path.join(dst, ROUND + "_headline_profile.md"), "w").write(md)
    print(md)
    other_kernels()


def other_kernels():
    """profiles/r02_other_kernels.md: the kernels of the side workloads (scripts/profile_side.sh)"""
    lines = [f"# {ROUND} — other kernels (1x MI355X; rocprofv3 --kernel-trace --stats of `bench.py --workload … --no-cpu-baseline --no-latency`)", ""]
    for wl, pat in (("recompute10m_graph", "beam_search_feat256_kernel"), ("recompute10m", "fused_fstat_kernel")):
        pj, pc = os.path.join(src, f"prof_{wl}.json"), os.path.join(src, f"prof_{wl}", ROUND + "_kernel_stats.csv")
        if not (os.path.exists(pj) and os.path.exists(pc)):
            continue
        j = json.load(open(pj))
        r = j["roofline"]
        rows = [x for x in csv.DictReader(open(pc)) if pat in x["Name"]]
        lines += [f"## {wl}: `{rows[0]['Name'][:70]}`" if rows else f"## {wl}", "",
                  f"* bench line of the profiled run: {j['value']:.0f} queries/s, recall@10 {j.get('recall_at_10')}, {r['bound']} {r['achieved']:.1f} {r['unit']} = {r['frac'] * 100:.1f} % of peak"]
        for x in rows:
            lines.append(f"* kernel-stats row `{x['Name'][:60]}`: Calls {x['Calls']}, AverageNs {float(x['AverageNs']):.0f}, MinNs {x['MinNs']}, MaxNs {x['MaxNs']}")
        if wl == "recompute10m_graph":
            pf, pw = (os.path.join(src, f"prof_rg_{c}", ROUND + "_counter_collection.csv") for c in ("FETCH_SIZE", "WRITE_SIZE"))
            if os.path.exists(pf) and os.path.exists(pw):
                try:
                    bf = json.load(open(os.path.join(src, "prof_rg_FETCH_SIZE.json")))
                    fe = sum(last_n(pf, "FETCH_SIZE", 4)) / 4 * 1024
                    wr = sum(last_n(pw, "WRITE_SIZE", 4)) / 4 * 1024
                    alg = bf["roofline"]["algorithmic_bytes_per_launch"]
                    json.dump({"workload": wl, "ef_search": bf["config"]["ef_search"], "kernel": "beam_search_feat256_kernel<1,4>",
                               "launch": "16384 queries, k=10, 10M x (256 bf16 features + f32 norm)", "round": ROUND, "fetch_bytes_raw": fe,
                               "fetch_bytes_corrected_x2_gfx950": 2 * fe, "write_bytes": wr, "hbm_bytes_per_launch": 2 * fe + wr,
                               "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (2 * fe + wr) / alg,
                               "method": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE), --kernel-include-regex on the query kernel, last 4 launches of "
                                         "`bench.py --workload recompute10m_graph --no-cpu-baseline --no-latency --steps 4 --warmup 1` (scripts/profile_side.sh); "
                                         "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM; split device layout (round 3): a 512-B feature row is four whole 128-B lines and the "
                                         "4-byte gather of its norm costs a fifth = 640 B per evaluated row, 1.24 x its 516 algorithmic bytes (the inline 520-B row of round 2 "
                                         "touched 5.06 lines = 648 B)"},
                              open(os.path.join(dst, f"pmc_traffic_{wl}.json"), "w"), indent=1)
                    lines.append(f"* PMC (separate passes): FETCH_SIZE x 2 + WRITE_SIZE = {(2 * fe + wr) / 1e9:.2f} GB per launch = {(2 * fe + wr) / alg:.3f} x the algorithmic "
                                 f"{alg / 1e9:.2f} GB (whole 128-B lines: four for the 512-B feature row + one for the 4-byte norm gather = 1.24 x) -> {(2 * fe + wr) / 1e9 / r['kernel_avg_ms']:.2f} TB/s of line traffic")
                except Exception as e:  # noqa: BLE001
                    lines.append(f"* PMC traffic pass: failed ({e})")
            lines.append(f"* bench.py HIP-event average of the timed launches: {r['kernel_avg_ms']:.3f} ms; algorithmic bytes per query {r['algorithmic_bytes_per_query']:.0f} "
                         f"({r['dist_evals_per_query']:.0f} evaluations x 516 B + {r['hops_per_query']:.0f} hops x 256 B); ceiling for random 520-B rows read four per instruction: 10 G rows/s = 5.2 TB/s algorithmic = 6.4 TB/s in whole 128-B lines (`r02_gather_ceiling.txt`)")
        lines.append("")
    pm = os.path.join(src, "prof_mfma", ROUND + "_counter_collection.csv")
    if os.path.exists(pm):
        busy, act = last_n(pm, "SQ_VALU_MFMA_BUSY_CYCLES", 1)[0], last_n(pm, "GRBM_GUI_ACTIVE", 1)[0]
        lines += ["## fused_fstat_kernel matrix-pipe utilisation (separate `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass, the 9.5M-row launch of a step)", "",
                  f"* SQ_VALU_MFMA_BUSY_CYCLES = {busy:.4e}, GRBM_GUI_ACTIVE = {act:.4e} summed over 8 XCDs -> MfmaUtil = busy / (active / 8 x 1024 SIMDs) = "
                  f"**{busy / (act / 8 * 1024) * 100:.1f} %** of the resident cycles (round 1: 72.6 %)", ""]
    pr = os.path.join(src, "prof_rdreq", ROUND + "_counter_collection.csv")
    if os.path.exists(pr):
        a1, a2 = sum(last_n(pr, "TCC_EA0_RDREQ_sum", 4)) / 4, sum(last_n(pr, "TCC_EA0_RDREQ_DRAM_sum", 4)) / 4
        lines += ["## hnsw10m: where the L2's read requests go (`--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum`)", "",
                  f"* {a1:.4e} requests per launch, {a2 / a1 * 100:.1f} % of them addressed to the DRAM path (x 128 B = {a1 * 128 / 1e9:.1f} GB = FETCH_SIZE x 2); the Infinity "
                  "Cache sits behind that interface and no counter of it is exposed on this pool", ""]
    fs = os.path.join(src, "forced_shard_rccl_world1.json")
    if os.path.exists(fs):
        open(os.path.join(dst, ROUND + "_forced_shard_rccl_world1.json"), "w").write([l for l in open(fs) if l.startswith("{")][-1])
        j = json.loads([l for l in open(fs) if l.startswith("{")][-1])
        lines += ["## The N > 1 code path of bench.py with one rank (`LEANN_BENCH_FORCE_SHARD=1 python -m torch.distributed.run --nproc-per-node 1 … bench.py --workload hnsw100k`)", "",
                  f"* shard mode through the library's RCCL group (ncclCommInitRank + ncclAllGather + merge kernel, tickets): {j['value']:.0f} queries/s, recall@10 {j['recall_at_10']:.4f}; "
                  f"`{j['config'].get('value_unit_note', '')}`", ""]
    open(os.path.join(dst, ROUND + "_other_kernels.md"), "w").write("\n".join(lines))


if __name__ == "__main__":
    main()
