"""bench.py's launch modes on the GPU box: the self-launching parent, and the N>1 code (sharded reads, chunked
all-gather, all-vs-all summary on the gathered matrix) with two ranks sharing the one GPU of the box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         env=e, timeout=900)
    assert out.returncode == 0, out.stderr.decode("utf-8", "replace")[-2000:]
    lines = out.stdout.decode().strip().splitlines()
    assert len(lines) == 1, lines  # ONE JSON line on stdout, nothing else
    return json.loads(lines[0])


def test_self_launch_relays_one_json_line():
    j = _bench(["--spawn", "--no-extras", "--no-cpu-baseline", "--reads", "20000", "--steps", "2", "--warmup", "1"])
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["value"] > 0 and j["roofline"]["bound"] == "hbm"


def test_config4_two_ranks_on_one_gpu():
    j = _bench(["--gpus", "2", "--reads", "40001", "--steps", "2", "--warmup", "1", "--queries", "64", "--ag-chunks", "3"],
               env={"KPOP_BENCH_SHARE_GPU": "1"})
    assert j["n_gpus"] == 2 and j["scaling"] == "strong"
    assert j["collective"]["ranks_in_communicator"] == 2 and j["collective"]["bytes_received_per_rank"] > 0
    assert j["gather_checksum_ok"] is True
    assert j["all_vs_all"]["every_query_finds_itself_at_distance_0"] is True and j["all_vs_all"]["against"] == 40001


def test_config4_through_rccl_world_size_1():
    j = _bench(["--workload", "config4", "--force-dist", "--reads", "30000", "--steps", "2", "--warmup", "1", "--queries", "32"])
    assert j["scaling"] == "strong" and j["collective"]["backend"].startswith("nccl") and j["gather_checksum_ok"] is True


def test_round_one_mode_still_runs_two_ranks():
    j = _bench(["--gpus", "2", "--workload", "headline", "--scaling", "weak", "--reads", "20000", "--steps", "2", "--warmup", "1"],
               env={"KPOP_BENCH_SHARE_GPU": "1"})
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["reads_per_gpu"] == 20000


@pytest.mark.parametrize("n", [1, 3])
def test_in_process_mode_through_the_c_abi(n):
    """`--in-process`: every device slot driven from one process through kpop_sharded_* (slots aliased to GPU 0 here)"""
    j = _bench(["--gpus", str(n), "--in-process", "--reads", "40001", "--steps", "2", "--warmup", "1", "--queries", "60", "--force-dist"],
               env={"KPOP_BENCH_SHARE_GPU": "1"})
    assert j["n_gpus"] == n and j["scaling"] == "strong" and j["value"] > 0
    assert j["launcher"].startswith("in-process")
    assert j["gather_checksum_ok"] is True
    assert len(j["per_device_ms"]["exposed_comm"]) == n
    assert j["all_vs_all"]["every_query_finds_itself_at_distance_0"] is True and j["all_vs_all"]["queries_total"] == 60 // n * n


def test_config4_reports_exposed_comm_and_startup():
    j = _bench(["--gpus", "2", "--reads", "40000", "--steps", "2", "--warmup", "1", "--queries", "16"], env={"KPOP_BENCH_SHARE_GPU": "1"})
    assert len(j["exposed_comm_ms_per_rank"]) == 2 and len(j["startup_s_per_rank"]) == 2
    assert "4 chunks" in j["config"]["workload"]  # --ag-chunks 0 = max(4, GPUs)
