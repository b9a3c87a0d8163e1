"""-m gpu: bench.py end to end on its smallest workload (hnsw100k: 100k x 768, seconds) — the contract's keys are there, the in-run checks
(GPU == oracle on the sampled queries; device hybrid rerank == oracle/searcher_oracle.py) hold, and the default-run orchestration merges its
legs into one line.  Guards the measurement harness itself; the numbers of record come from the full-size workloads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "stdout must carry exactly one JSON line"
    return json.loads(lines[0])


def test_bench_line_contract_and_in_run_identity_check(gpu):
    j = _bench("--workload", "hnsw100k", "--steps", "3", "--warmup", "1", "--no-latency", "--cpu-queries", "512", "--small-batch", "64")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "recall_at_10"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["vs_baseline"] is None and j["dtype"] == "f32"
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert j["recall_at_10"] >= 0.95
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["gpu_results_bit_identical_on_sample"] is True
    assert c["simd_value"] > 0
    sb = j["small_batch"]
    assert sb["batch"] == 64 and sb["ms_per_call"] > 0
    # independent small calls overlap (8 streams), and the 8 calls replay from ONE HIP graph: the device entry points are capturable
    assert sb["value_concurrent"] > sb["value"]
    if sb["value_concurrent_hipgraph"] is None:  # capture is an extra of the harness (bench.py logs why it failed); never observed on this pool
        print("note: HIP-graph capture of the small-batch calls was not available in this run")
    else:
        assert sb["value_concurrent_hipgraph"] > sb["value"]


def test_bench_hybrid_leg_checks_itself_against_the_oracle(gpu):
    j = _bench("--workload", "hnsw100k", "--hybrid", "--steps", "3", "--warmup", "1", "--no-latency", "--cpu-queries", "1024", "--compat-polarity", "false")
    h = j["hybrid"]
    assert h["fetch_k"] == 50 and h["compat_polarity"] is False and h["rerank_avg_ms"] > 0
    assert h["rerank_parity"]["sample"] == 1024 and h["rerank_parity"]["mismatching_queries"] == 0
    assert h["cpu_port_equals_numpy_restatement"] is True
    assert j["cpu_baseline"]["gpu_results_bit_identical_on_sample"] is True


def test_gpus_2_rehearsals_on_one_gpu(gpu):
    """the N > 1 paths with what one GPU allows: two ranks over gloo sharing the device (started by bench.py itself), and two shards behind
    one composite handle"""
    j = _bench("--gpus", "2", "--workload", "hnsw100k", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-latency",
               env={"LEANN_BENCH_DIST_BACKEND": "gloo"})
    assert j["n_gpus"] == 2 and "shard2" in j["config"]["parallelism"] and j["recall_at_10"] >= 0.95 and "replica_mode" in j
    j = _bench("--gpus", "2", "--mode", "composite", "--workload", "hnsw100k", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-latency",
               env={"LEANN_BENCH_COMPOSITE_DEVICES": "0,0"})
    assert j["n_gpus"] == 2 and j["config"]["parallelism"].startswith("composite2") and j["recall_at_10"] >= 0.95
    assert j["scaling"] == "weak" and j["config"]["corpus_rows_total"] == 200_000
    # --strong: the workload's rows are the whole corpus, split over the shards
    j = _bench("--gpus", "2", "--mode", "composite", "--strong", "--workload", "hnsw100k", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-latency",
               env={"LEANN_BENCH_COMPOSITE_DEVICES": "0,0"})
    assert j["scaling"] == "strong" and j["config"]["rows_per_gpu"] == 49_984 and j["config"]["corpus_rows_total"] == 2 * 49_984


def test_default_run_parent_needs_no_gpu_and_relays_one_line(gpu):
    """the driver's command shape (no --workload): the parent only starts children — here the headline leg alone (LEANN_BENCH_SKIP_OTHERS) at
    2 steps — and prints ONE line with the `other_configs` key"""
    j = _bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-latency", env={"LEANN_BENCH_SKIP_OTHERS": "1"})
    assert j["steps"] == 2 and j["warmup"] == 1 and j["n_gpus"] == 1 and j["config"]["rows_per_gpu"] == 10_000_000
    assert j["other_configs"] == {} and j["recall_at_10"] >= 0.95 and 0.5 < j["roofline"]["frac"] < 1
