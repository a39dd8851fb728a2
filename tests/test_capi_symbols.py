"""CPU: libmiso_amd.so loads and exports every function include/miso_amd.h declares; with no GPU
the sampler entry points fail loudly (there is no CPU fallback to fall into)."""
import ctypes
import os
import sys
import re

import numpy as np
import pytest

import miso_amd
from miso_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# the two headers that declare exported functions (miso_detmath.h / miso_philox.h are inline-only)
API_HEADERS = ("miso_amd.h", "miso_alnio.h")


def declared_functions():
    names = set()
    for h in API_HEADERS:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(miso_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_only_inline_headers_are_left_out():
    for h in os.listdir(os.path.join(ROOT, "include")):
        if h not in API_HEADERS:
            src = open(os.path.join(ROOT, "include", h)).read()
            assert "extern \"C\"" not in src, h


def test_header_declares_the_path():
    names = declared_functions()
    for must in ("miso_create_gene", "miso_run", "miso_run_paired", "miso_batch_create",
                 "miso_batch_add_event", "miso_batch_launch", "miso_batch_get_result",
                 "miso_match_iso", "miso_match_iso_paired", "miso_last_error", "miso_aln_open",
                 "miso_aln_fetch", "miso_aln_parse_reads"):
        assert must in names


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(capi.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_library_and_checker_draw_by_the_same_contract_version():
    """two rounds changed seeded counter-mode outputs without a marker.  The library and the CPU checker are
    compiled from one header (include/miso_philox.h MISO_CONTRACT_VERSION); a stale prebuilt checker, or results saved by
    another version, now show up here instead of as a parity failure nobody can explain."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import OrcLib
    src = open(os.path.join(ROOT, "include", "miso_philox.h")).read()
    want = int(re.search(r"#define MISO_CONTRACT_VERSION (\d+)", src).group(1))
    assert capi.contract_version() == want
    assert OrcLib().contract_version() == want


def test_no_cpu_fallback_without_device():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    g = miso_amd.Gene([(1, 100), (201, 300), (401, 500)], [[0, 1, 2], [0, 2]])
    b = miso_amd.Batch(36, iters=50, burn=10, lag=1, chains=1)
    b.add_event(g, np.array([10, 250], np.int32), [b"36M", b"36M"])
    with pytest.raises(miso_amd.InternalError, match="no HIP device"):
        b.run()
    samples = np.zeros(80)
    rc = capi.lib().miso_run(g.handle, capi._p(np.array([10], np.int32)), capi._cigs([b"36M"]), 1,
                             36, 1, 1, 50, 100000, 10, 1, capi._p(np.ones(2)), 2, 0, 0, 0,
                             ctypes.c_uint64(1), capi._p(samples), None, None, None, None, None, None)
    assert rc == capi.MISO_ENODEVICE
    assert b"no HIP device" in capi.lib().miso_last_error()


def test_product_does_not_reference_the_oracle():
    """The product tree must not import, link or name anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "miso_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", ".c")) or f == "Makefile":
                txt = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"oracle/|miso_oracle|libmiso_oracle|libmiso_ref|_libs", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
