"""`python bench.py --gpus N` as the driver invokes it (no launcher around it): bench.py starts the N ranks itself as a child process,
relays rank 0's JSON line and returns the children's exit code; when the ranks produce no line it tries the one-process composite
handle.  No GPU needed: the child command is intercepted."""
import importlib.util
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launch_command_is_one_rank_per_gpu_under_torch_distributed_run():
    b = _bench()
    cmd = b.launch_command(8, 29511, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    comp = b.composite_command(["--gpus", "8", "--mode", "shard", "--steps", "3"])
    assert comp == [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--mode", "composite"]


def test_self_launch_relays_the_line_and_falls_back_to_the_composite_handle(capsys):
    b = _bench()
    args = types.SimpleNamespace(gpus=4, workload="hnsw100k", mode="shard")
    seen = []

    def ok(cmd):
        seen.append(cmd)
        return types.SimpleNamespace(returncode=0, stdout=b'NCCL banner\n{"n_gpus": 4, "value": 1.0}\n')
    assert b.self_launch(args, ["--gpus", "4"], run=ok) == 0
    assert capsys.readouterr().out.strip() == '{"n_gpus": 4, "value": 1.0}'
    assert len(seen) == 1 and "--nproc-per-node=4" in seen[0]

    seen.clear()

    def ranks_fail(cmd):
        seen.append(cmd)
        if "torch.distributed.run" in cmd:
            return types.SimpleNamespace(returncode=1, stdout=b"")
        return types.SimpleNamespace(returncode=0, stdout=b'{"n_gpus": 4, "config": {"parallelism": "composite4"}}\n')
    assert b.self_launch(args, ["--gpus", "4"], run=ranks_fail) == 0
    assert len(seen) == 2 and seen[1][-2:] == ["--mode", "composite"]
    assert "composite4" in capsys.readouterr().out

    def all_fail(cmd):
        return types.SimpleNamespace(returncode=3, stdout=b"")
    assert b.self_launch(args, ["--gpus", "4"], run=all_fail) == 3


def test_gpus_2_without_a_gpu_fails_in_the_child_ranks_not_in_an_argument_check():
    """here (no GPU) the ranks start, find no device and say so; the parent returns their failure"""
    env = dict(os.environ, LEANN_BENCH_NO_COMPOSITE_FALLBACK="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "hnsw100k", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = p.stderr.decode(errors="replace")
    assert p.returncode != 0
    assert "torch.distributed.run" in err  # the spawned command line is logged
    assert "needs a GPU" in err or "GPUs requested" in err  # said by the ranks
    assert "must be launched" not in err


def test_default_run_reports_every_one_gpu_config_in_one_line(capsys, monkeypatch):
    """`python bench.py --gpus 1 --steps K --warmup W` (the driver's shape): headline child first (with the caller's flags +
    --headline-only), then one child per other 1-GPU BASELINE config; ONE line comes out, the headline's plus `other_configs`."""
    import json
    b = _bench()
    args = types.SimpleNamespace(gpus=1)
    seen = []

    def run(cmd):
        seen.append(cmd)
        if "--headline-only" in cmd:
            return types.SimpleNamespace(returncode=0, stdout=b'{"metric": "m", "value": 1.0, "config": {"workload": "hnsw10m"}}\n')
        wl = cmd[cmd.index("--workload") + 1]
        if wl == "recompute10m":
            return types.SimpleNamespace(returncode=3, stdout=b"")
        line = {"value": 2.0, "unit": "queries/s", "recall_at_10": 0.97, "config": {"workload": wl, "ef_search": 128},
                "roofline": {"bound": "hbm", "frac": 0.7, "kernel": "k", "kernel_avg_ms": 1.5}}
        return types.SimpleNamespace(returncode=0, stdout=(json.dumps(line) + "\n").encode())
    assert b.run_all_configs(args, ["--gpus", "1", "--steps", "20", "--warmup", "5"], run=run) == 0
    line = json.loads(capsys.readouterr().out.strip())
    assert seen[0][-5:] == ["--steps", "20", "--warmup", "5", "--headline-only"]
    assert line["value"] == 1.0 and set(line["other_configs"]) == {n for n, _, _ in b.OTHER_CONFIGS}
    assert line["other_configs"]["hnsw1m_ef128"]["roofline"]["frac"] == 0.7 and line["other_configs"]["hnsw1m_ef128"]["kernel_avg_ms"] == 1.5
    assert "failed" in line["other_configs"]["recompute10m_exhaustive_batch64"]
    # a failing headline fails the run; nothing else is started
    seen.clear()
    assert b.run_all_configs(args, [], run=lambda cmd: (seen.append(cmd), types.SimpleNamespace(returncode=2, stdout=b""))[1]) == 2
    assert len(seen) == 1
    # out of budget: legs are named as skipped, not silently dropped
    monkeypatch.setenv("LEANN_BENCH_BUDGET_S", "1")
    capsys.readouterr()
    assert b.run_all_configs(args, [], run=run) == 0
    line = json.loads(capsys.readouterr().out.strip())
    assert all("skipped" in v for v in line["other_configs"].values())


def test_hanging_ranks_are_ended_at_the_deadline_and_the_fallback_runs(capsys, monkeypatch):
    """a rank that never returns (an RCCL bootstrap that hangs) must not eat the run: past LEANN_BENCH_LAUNCH_TIMEOUT_S the child's own
    process group is ended and the composite fallback gets its turn"""
    import time
    b = _bench()
    monkeypatch.setenv("LEANN_BENCH_LAUNCH_TIMEOUT_S", "2")
    monkeypatch.setattr(b, "launch_command", lambda n, port, argv: [sys.executable, "-c", "import time; time.sleep(120)", "torch.distributed.run"])
    monkeypatch.setattr(b, "composite_command", lambda argv: [sys.executable, "-c", "print('{\"n_gpus\": 2, \"config\": {\"parallelism\": \"composite2\"}}')"])
    args = types.SimpleNamespace(gpus=2, workload="hnsw100k", mode="shard")
    t0 = time.time()
    assert b.self_launch(args, ["--gpus", "2"]) == 0
    assert time.time() - t0 < 40
    assert "composite2" in capsys.readouterr().out
