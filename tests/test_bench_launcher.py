"""bench.py's launcher (no GPU needed: the parent never touches one, and the test switch stops the ranks before they would).
The first N > 1 run happens on hardware this repo has never seen: a rank that hangs at the rendezvous must become a bounded,
explained failure -- the watchdog's exit code, every rank's last lines -- not a silent driver timeout."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *argv, timeout=120):
    env = dict(os.environ, **extra_env)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode(), time.time() - t0


def test_hanging_ranks_end_in_the_watchdogs_exit_code_and_tails():
    rc, out, err, took = _bench({"KPOP_BENCH_FAKE_HANG": "all", "KPOP_BENCH_NO_FALLBACK": "1"}, "--gpus", "2", "--timeout", "10")
    assert rc == 124, (rc, err[-2000:])
    assert took < 60, took
    assert out == ""  # no JSON line was made up
    assert "no rank finished within --timeout 10 s" in err
    for r in (0, 1):  # every rank's own stderr, kept apart and printed by the launcher
        assert "---- rank%d.stderr (last 40 lines)" % r in err
        assert "rank %d: KPOP_BENCH_FAKE_HANG" % r in err


def test_one_hanging_rank_is_named():
    # rank 1 never joins; rank 0 gets as far as it can on this box (no GPU here: it says so and exits; on a GPU box it would
    # wait at the rendezvous until the watchdog).  Either way the launcher ends non-zero within the bound and shows rank 1's line.
    rc, out, err, took = _bench({"KPOP_BENCH_FAKE_HANG": "1", "KPOP_BENCH_NO_FALLBACK": "1"}, "--gpus", "2", "--timeout", "20")
    assert rc != 0 and took < 90, (rc, took)
    assert out == ""
    assert "rank 1: KPOP_BENCH_FAKE_HANG" in err


def test_failed_ranks_are_retried_in_process_and_the_retry_is_reported():
    # without a GPU both attempts fail; what is checked is that the second one was made, from a fresh child, and said why
    rc, out, err, took = _bench({"KPOP_BENCH_FAKE_HANG": "all"}, "--gpus", "2", "--timeout", "8")
    assert rc != 0 and took < 90, (rc, took)
    assert "trying the same job --in-process" in err
    assert "the --in-process attempt" in err


def test_config5_goes_through_the_same_launcher_and_fall_back():
    # `--workload config5 --gpus 2`: the ranks hang (the test switch), the watchdog ends them, the in-process form of the SAME workload is
    # tried from a fresh child (here it fails for want of a GPU: what is checked is that it was the config-5 job that was retried)
    rc, out, err, took = _bench({"KPOP_BENCH_FAKE_HANG": "all"}, "--gpus", "2", "--workload", "config5", "-k", "13", "--timeout", "8")
    assert rc != 0 and took < 90, (rc, took)
    assert out == ""
    assert "trying the same job --in-process" in err and "the --in-process attempt" in err


def test_config5_shape_and_line_are_built_without_a_gpu():
    """the workload's defaults (k = 15, D = 16, 10,000 reads unless the command line says otherwise) and the line's fixed fields"""
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--workload", "config5", "--gpus", "4"])
    assert bench._config5_shape(a) == (15, 16, 10000, 150)
    a = bench.parse_args(["--workload", "config5", "-k", "13", "--dims", "8", "--reads", "500", "--steps", "2"])
    assert bench._config5_shape(a) == (13, 8, 500, 150)
    line = bench._config5_line(a, 2, 13, 8, 500, 150, 0.004, "launcher", "sharding", {"n_ranks_seen": 2})
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["dtype"] == "f64" and line["n_ranks_seen"] == 2
    assert abs(line["value"] - 500 * 2 / 0.004) < 1e-6 and abs(line["ms_per_step"] - 2.0) < 1e-9
    assert "config 5" in line["config"]["workload"] and "33554432 canonical" in line["config"]["workload"]
