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


def test_default_line_carries_every_baseline_config():
    """the one-GPU line the driver records: BASELINE configs 2, 3 and 5 as legs of their own (config 4's N = 1 point was there),
    each with ms_per_step and a roofline against a bound that can hold it (no fraction above 1); all-vs-all timed warm"""
    j = _bench(["--no-children", "--cpu-seconds", "2", "--steps", "5", "--warmup", "2"])
    assert j["n_gpus"] == 1 and 0.0 < j["roofline"]["frac"] <= 1.0 and j["cpu_baseline"]["kind"] == "port"
    c2, c3, c5 = j["config2_on_this_gpu"], j["config3_on_this_gpu"], j["config5_on_this_gpu"]
    assert c2["ms_per_step"] > 0 and c2["roofline"]["bound"] == "infinity_cache" and 0.0 < c2["roofline"]["frac"] <= 1.0
    for leg, bound in (("unrelated_genomes", "hbm"), ("one_organism_0.3pct", "mfma")):
        r = c3[leg]["roofline"]
        assert c3[leg]["ms_per_step"] > 0 and r["bound"] == bound and 0.0 < r["frac"] <= 1.0, (leg, r)
    assert c3["one_organism_0.3pct"]["speedup_over_streaming_kernel"] > 1.5  # (the tile route is the default)
    assert "skipped" in c5 or (c5["rows_finite_and_inside_the_coefficient_range"] and 0.0 < c5["roofline"]["frac"] <= 1.0)
    ava = j["config4_on_this_gpu"]["all_vs_all"]
    assert ava["every_query_finds_itself_at_distance_0"] and ava["first_call_seconds"] > 0 and ava["seconds"] > 0


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


def test_config5_the_twisters_rows_over_two_ranks_and_over_two_slots():
    """`--workload config5` (k-mer rows of the twister sharded, ONE all-reduce of the partial rows) at a k the box holds twice: one
    GPU whole, two ranks sharing the GPU (gloo through the host), two device slots from one process (--in-process: the launcher's
    fall-back) -- the same rows to 1e-12 whichever way, both ranks seen"""
    common = ["--workload", "config5", "-k", "13", "--dims", "16", "--reads", "4000", "--steps", "2", "--warmup", "1"]
    one = _bench(["--gpus", "1"] + common)
    two = _bench(["--gpus", "2"] + common, env={"KPOP_BENCH_SHARE_GPU": "1"})
    slots = _bench(["--gpus", "2", "--in-process"] + common, env={"KPOP_BENCH_SHARE_GPU": "1"})
    assert one["n_gpus"] == 1 and one["n_ranks_seen"] == 1 and 0.0 < one["roofline"]["frac"] <= 1.0
    assert two["n_gpus"] == 2 and two["n_ranks_seen"] == 2 and slots["n_ranks_seen"] == 2 and slots["launcher"].startswith("in-process")
    assert len(two["per_rank"]["windows_of_own_kmers"]) == 2 and abs(sum(two["per_rank"]["windows_of_own_kmers"]) - 4000 * (150 - 13 + 1)) < 0.5
    for j in (one, two, slots):
        assert j["rows_finite_and_inside_the_coefficient_range"] is True and "config 5" in j["config"]["workload"]
    a = one["rows_digest"]
    for j in (two, slots):
        b = j["rows_digest"]
        assert abs(a["sum"] - b["sum"]) <= 1e-12 * max(1.0, abs(a["sum"])) * 100 and abs(a["sum_of_squares"] - b["sum_of_squares"]) <= 1e-12 * a["sum_of_squares"] * 100
        for x, y in zip(a["row0"] + a["row_last"], b["row0"] + b["row_last"]):
            assert abs(x - y) <= 1e-12 * max(abs(x), 1e-3)


def test_config5_through_rccl_world_size_1():
    """the sharded form of config 5 with ONE rank (--force-dist): the slice is the whole hash space + the all-ones column, the partial rows go
    through RCCL's all-reduce (world size 1: the collective path without a second GPU) and come back divided by the reduced acc -- the rows of
    the plain one-GPU form to 1e-12"""
    common = ["--workload", "config5", "-k", "13", "--dims", "16", "--reads", "4000", "--steps", "2", "--warmup", "1", "--gpus", "1"]
    plain = _bench(common)
    forced = _bench(common + ["--force-dist"])
    assert forced["launcher"].startswith("torch.distributed") and "RCCL" in forced["launcher"] and forced["n_ranks_seen"] == 1
    assert "all-reduce" in forced["config"]["sharding"] and forced["rows_finite_and_inside_the_coefficient_range"] is True
    a, b = plain["rows_digest"], forced["rows_digest"]
    assert abs(a["sum"] - b["sum"]) <= 1e-10 * max(1.0, abs(a["sum"])) and abs(a["sum_of_squares"] - b["sum_of_squares"]) <= 1e-10 * a["sum_of_squares"]
    for x, y in zip(a["row0"] + a["row_last"], b["row0"] + b["row_last"]):
        assert abs(x - y) <= 1e-12 * max(abs(x), 1e-3)
