"""CPU tests of the C-ABI library: it loads, exports every symbol the header declares, its
host-only helpers agree with the oracle, and compute entry points fail loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from dipper_amd import capi
    return capi.load_library()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "dipper_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(dpr_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    for nm in names:
        assert hasattr(lib, nm), f"libdipper_hip.so does not export {nm}"
    assert lib.dpr_abi_version() == 1


def test_packers_match_oracle(orc):
    from dipper_amd import capi
    rng = np.random.default_rng(0)
    alphabet = np.frombuffer(b"ACGTUNacgtu-*XRY", dtype=np.uint8)
    for L in (0, 1, 15, 16, 17, 31, 32, 33, 64, 1000, 4097):
        s = alphabet[rng.integers(0, len(alphabet), size=L)].tobytes()
        assert np.array_equal(capi.pack4(s), orc.pack4(s))
        assert np.array_equal(capi.pack2(s), orc.pack2(s))


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_helpers_are_consistent(lib, world):
    n = 1000
    seen = np.zeros(n, dtype=np.int64)
    for r in range(world):
        cnt = lib.dpr_shard_rows(n, r, world)
        owned = [i for i in range(n) if lib.dpr_shard_owner(i, world) == r]
        assert cnt == len(owned)
        for k, i in enumerate(owned):
            assert lib.dpr_shard_local_row(i, world) == k           # local storage is dense, in order
            assert lib.dpr_shard_global_row(k, r, world) == i
            seen[i] += 1
    assert np.all(seen == 1)
    # active-size monotonicity: slots < n' of a rank are a prefix of its local rows
    for npr in (1, 63, 64, 65, 500):
        assert sum(lib.dpr_shard_rows(npr, r, world) for r in range(world)) == npr


def test_nj_key_matches_literal_emulation(lib, orc):
    from tests import _util
    rng = np.random.default_rng(9)
    n = 300
    D = rng.integers(1, 3, size=(n, n)).astype(np.float64)
    D = np.tril(D, -1) + np.tril(D, -1).T
    U = orc.row_sums(np.ascontiguousarray(D))
    x, y, q = _util.ref_findmin_emulation(D, U, n)
    # brute force with the library's key
    r = float(n - 2)
    best = (np.inf, 2**64)
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            qq = D[i, j] - U[i] / r - U[j] / r
            if qq == q:
                best = min(best, (qq, lib.dpr_nj_key(i, j, n)))
    key = best[1]
    assert (key & 0xFFFFFF, (key >> 24) & 0xFFFFFF) == (x, y)


def test_record_reduce(lib):
    rec = np.zeros(4, dtype=[("q", "f8"), ("key", "u8"), ("d", "f8"), ("pad", "u8")])
    rec["q"] = [1.0, -2.0, -2.0, 10000.0]
    rec["key"] = [5, 9, 7, 2**64 - 1]
    assert lib.dpr_record_reduce(rec.ctypes.data, 4) == 2
    rec["key"] = 2**64 - 1
    assert lib.dpr_record_reduce(rec.ctypes.data, 4) == -1


def test_no_cpu_fallback():
    """Without a GPU the compute path must fail loudly, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import dipper_amd
    with pytest.raises(dipper_amd.DipperError) as ei:
        dipper_amd.Dipper(0)
    assert "no HIP device" in str(ei.value) or "Gpu_ERROR" in str(ei.value)


def test_product_does_not_reference_oracle():
    """The product tree must not import, link or call anything under oracle/."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "dipper_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle|_orc\b|liboracle", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
