"""bench.py runs the row-sharded legs of several ranks in child processes without torch (njs_worker) and talks to them in JSON
lines with time limits.  On a host without a GPU the child cannot create its context: what must come back is ONE error line,
promptly, and a closed child -- the path every failure of a leg takes (a parent never waits without a limit)."""
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_child_reports_failure_as_a_line_and_ends():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU (the child would run the leg)")
    sys.path.insert(0, ROOT)
    import bench
    t0 = time.time()
    ch = bench.Child({"rank": 0, "world": 2, "device": 0, "tips": 64, "sites": 64, "iters": 4, "p4": os.devnull, "plan": "mailbox", "local": True})
    try:
        m = ch.get(60.0)
        assert isinstance(m, dict) and "error" in m, m
        m2 = ch.get(10.0)                      # nothing else follows: end of file, reported as an error too
        assert "error" in m2
    finally:
        ch.close()
    assert ch.p.poll() is not None
    assert time.time() - t0 < 60


def test_child_silence_is_bounded():
    """a child that says nothing is reported after the limit and killed by close()"""
    sys.path.insert(0, ROOT)
    import bench
    import subprocess
    ch = bench.Child.__new__(bench.Child)
    import queue
    import threading
    ch.q = queue.Queue()
    ch.p = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    threading.Thread(target=ch._pump, daemon=True).start()
    t0 = time.time()
    m = ch.get(1.5)
    assert "error" in m and time.time() - t0 < 10
    ch.p.kill()
    ch.close()
    assert ch.p.poll() is not None


def test_scaling_summary_sits_behind_roofline_and_labels_rccl():
    """the compact `nj_iteration_scaling` object (north_star's scaling metric) comes right behind `roofline` in the bench line,
    carries `ranks` and `rccl` separately (a run joined without RCCL must not be reported as rccl_ranks) and leaves the contract's
    `scaling` string alone"""
    sys.path.insert(0, ROOT)
    import bench
    out = {"metric": "m", "value": 1.0, "scaling": "weak", "roofline": {"frac": 0.7}, "parity_check": {}, "nj_scaling": {
        "tips": 30000, "iterations_timed": 256, "streaming_one_gpu": {"nj_iterations_per_s": 1000.0}, "default_plan_one_gpu": {"nj_iterations_per_s": 60000.0},
        "row_sharded": {"legacy": {"nj_iterations_per_s": 3000.0, "iteration_speedup_vs_one_gpu": 3.0, "ranks": 8, "rccl": True, "rccl_ranks": 8,
                                   "matches_single_gpu": True, "launches_per_iteration": 4.0, "collectives_per_iteration": 2.0},
                        "mailbox": {"nj_iterations_per_s": 5000.0, "iteration_speedup_vs_one_gpu": 5.0, "ranks": 2, "rccl": False, "matches_single_gpu": True,
                                    "launches_per_iteration": 2.0, "collectives_per_iteration": 0.0},
                        "peer": {"error": "child said nothing for 30 s (killed)"}}}}
    new = bench.with_scaling_summary(dict(out), 8)
    keys = list(new)
    assert keys.index("nj_iteration_scaling") == keys.index("roofline") + 1
    assert new["scaling"] == "weak"
    comp = new["nj_iteration_scaling"]
    assert comp["streaming_one_gpu_its_per_s"] == 1000.0 and comp["default_plan_one_gpu_its_per_s"] == 60000.0 and comp["n_gpus"] == 8
    assert comp["row_sharded"]["legacy"] == {"its_per_s": 3000.0, "speedup_vs_streaming_one_gpu": 3.0, "ranks": 8, "rccl": True, "matches_single_gpu": True,
                                             "launches_per_iteration": 4.0, "collectives_per_iteration": 2.0}
    assert comp["row_sharded"]["mailbox"]["rccl"] is False and comp["row_sharded"]["mailbox"]["ranks"] == 2
    assert comp["row_sharded"]["peer"] == {"error": "child said nothing for 30 s (killed)"}
    # no nj_scaling leg (skipped by the budget): the line is returned unchanged
    assert bench.with_scaling_summary({"roofline": {}, "nj_scaling": {"skipped": "budget"}}, 1) == {"roofline": {}, "nj_scaling": {"skipped": "budget"}}


def test_cli_phase_lines_are_parsed():
    """bench.py reads the command's own progress lines (with and without a colon before the number)"""
    sys.path.insert(0, ROOT)
    import bench
    text = ("Read 550000 sequences from input file.\nTree loaded successfully with 999999 nodes and root node_550000.\nInput in: 132 ms\n"
            "Device ready in: 111 ms\nSketch Created in: 38 ms\nDistance Operation Time 3640 ms\nTree Operation Time 2269 ms\n"
            "Distance batches overlapped with tree operations: 0 of 49, 0 ms in flight\nTree Created in: 557 ms\n")
    assert bench.cli_phases(text) == {"input_ms": 132.0, "device_ready_ms": 111.0, "sketch_ms": 38.0, "distance_ms": 3640.0, "tree_op_ms": 2269.0, "tree_ms": 557.0}
