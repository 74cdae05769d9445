"""The C-ABI shared library loads on a GPU-less host, exports every symbol
include/mmgp.h declares, carries a gfx950 code object, and refuses to compute
without a device (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from tests import helpers as H

ROOT = H.ROOT


def _lib_path():
    from meshlessmultigridpoisson_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        pytest.skip("libmmgp.so not built (run __graft_entry__.build())")
    return _capi.LIB_PATH


def test_header_symbols_are_exported():
    from meshlessmultigridpoisson_amd import _capi
    L = C.CDLL(_lib_path())
    header = open(os.path.join(ROOT, "include", "mmgp.h")).read()
    declared = sorted(set(re.findall(r"\b(mmg_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in mmgp.h but not exported"
    assert sorted(_capi.SYMBOLS) == declared, "python binding list out of sync with mmgp.h"


def test_library_contains_gfx950_code_object():
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = shutil.copy(_lib_path(), d)
        out = subprocess.run([objdump, "--offloading", p], capture_output=True, text=True, cwd=d).stdout
    assert "gfx950" in out and "gfx906" not in out, out


def test_no_cpu_fallback_without_device():
    from meshlessmultigridpoisson_amd import _capi
    if _capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    case = H.load_case("dirichlet_3level")
    with pytest.raises(_capi.MmgError, match="no HIP device"):
        H.device_level(H.level_arrays(case, 0))
    h = C.c_void_p()
    rc = _capi.lib().mmg_transfer_create(C.byref(h), 1, 1, None, None, None, 0)
    assert rc != 0


def test_host_library_loads_and_links_capi():
    from meshlessmultigridpoisson_amd import _host
    L = _host.lib()
    for name in ("mmgh_mg_create_square", "mmgh_grid_sor", "mmgh_points_from_msh", "mmgh_mg_vcycle"):
        assert hasattr(L, name)


def test_auto_tile_points_is_device_free_arithmetic():
    from meshlessmultigridpoisson_amd import _capi
    # device-filling level: largest multiple of 256 that keeps 4 wavefronts per CU (single dependency-driven launch)
    assert _capi.auto_tile_points(10077696, 3, 50, 2, 256, 163840) == 1280
    assert _capi.auto_tile_points(10077696, 3, 50, 0, 256, 163840) == 1280  # lanes 0: as mmg_level_create picks L
    # latency-bound levels take the dense multi-wavefront layout (capi.hip: level_layout): its tile sizes
    assert _capi.auto_tile_points(2000000, 3, 50, 2, 256, 163840) == 1024
    assert _capi.auto_tile_points(108 ** 3, 3, 50, 0, 256, 163840) == 512
    assert _capi.auto_tile_points(54 ** 3, 3, 50, 0, 256, 163840) == 256
    assert _capi.auto_tile_points(250000, 2, 25, 0, 256, 163840) == 256
    assert _capi.auto_tile_points(1000000, 2, 37, 0, 256, 163840) == 512
    assert _capi.auto_tile_points(4000000, 2, 37, 4, 256, 163840) >= 256     # packed stream beyond the dense regime
    # 4e6 ... 7.5e6 points (round 3): dense layout with the extra entry plane, 1024-point tiles (measured 171^3: 65 %
    # against 57 % for the packed stream with 384-point tiles; 190^3: 67 against 63 %); beyond it the packed stream
    assert _capi.auto_tile_points(171 ** 3, 3, 50, 2, 256, 163840) == 1024
    assert _capi.auto_tile_points(190 ** 3, 3, 50, 0, 256, 163840) == 1024
    assert _capi.auto_tile_points(200 ** 3, 3, 50, 0, 256, 163840) in (384, 1280)
    assert _capi.auto_tile_points(10000, 2, 37, 4, 256, 163840) == 256


def test_host_threads_follow_the_override_and_the_cpu_share(monkeypatch):
    """mmg_host_threads: MMG_NUM_THREADS when set (read on every call: bench.py sets it per rank after the library
    is loaded), else at most the CPUs of the affinity mask (capped by the container's quota, which cannot exceed it)."""
    import ctypes
    import os
    from meshlessmultigridpoisson_amd import _capi
    f = _capi.lib().mmg_host_threads
    f.restype = ctypes.c_int
    monkeypatch.delenv("MMG_NUM_THREADS", raising=False)
    auto = f()
    assert 1 <= auto <= len(os.sched_getaffinity(0))
    monkeypatch.setenv("MMG_NUM_THREADS", "3")
    assert f() == 3
    monkeypatch.delenv("MMG_NUM_THREADS")
    assert f() == auto
