"""CPU tests of the `dipper` command line: flag handling, error behaviour and input parsing that
happen before the GPU is touched (src/tree_generation.cu:159-188,593-599)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "dipper_amd", "bin", "dipper")


def run(*args):
    return subprocess.run([BIN, *args], capture_output=True, text=True)


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        g.build()


def test_help_goes_to_stderr_exit_0():
    r = run("-h")
    assert r.returncode == 0 and "DIPPER Command Line Arguments" in r.stderr and r.stdout == ""
    for flag in ("--input-format", "--input-file", "--output-file", "--output-format", "--algorithm",
                 "--placement-mode", "--kmer-size", "--sketch-size", "--distance-type", "--add", "--input-tree"):
        assert flag in r.stderr


def test_missing_required_is_red_and_exit_1(tmp_path):
    r = run("-i", "d", "-O", str(tmp_path / "o.nwk"))
    assert r.returncode == 1 and "\033[31m" in r.stderr and "--input-file" in r.stderr


def test_add_requires_tree(tmp_path):
    r = run("-i", "m", "-I", "x.fa", "-O", str(tmp_path / "o"), "--add")
    assert r.returncode == 1 and "--input-tree/-t" in r.stderr


def test_unknown_option():
    r = run("--bogus")
    assert r.returncode == 1 and "\033[31m" in r.stderr


def test_missing_input_file(tmp_path):
    r = run("-i", "d", "-I", str(tmp_path / "nope.phy"), "-O", str(tmp_path / "o"))
    assert r.returncode == 1 and "Cannot open file" in r.stderr
    r = run("-i", "m", "-I", str(tmp_path / "nope.fa"), "-O", str(tmp_path / "o"))
    assert r.returncode == 1 and "cant open file" in r.stderr


def test_invalid_combination(tmp_path):
    p = tmp_path / "a.fa"
    p.write_text(">a\nACGT\n>b\nACGA\n")
    r = run("-i", "x", "-I", str(p), "-O", str(tmp_path / "o"))
    assert r.returncode == 1 and "Invalid input-output combinations" in r.stdout


def test_truncated_phylip(tmp_path):
    p = tmp_path / "bad.phy"
    p.write_text("3\nA\nB\t0.1\nC\t0.2\n")
    r = run("-i", "d", "-I", str(p), "-O", str(tmp_path / "o"))
    assert r.returncode == 1 and "PHYLIP row 2" in r.stderr
