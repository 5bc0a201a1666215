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
