"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, and exports exactly
what include/echopype_amd.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "echopype_amd.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(epa_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for must in ("epa_power_coef_ek", "epa_sv_power", "epa_sv_mvbs_fused", "epa_mvbs", "epa_mvbs_index",
                 "epa_noise_estimate", "epa_noise_apply", "epa_sv_complex", "epa_time_bin_offsets",
                 "epa_last_error", "epa_nanminmax"):
        assert must in syms


def test_library_builds_and_exports_every_declared_symbol():
    from echopype_amd.build import LIBPATH, build_library

    path = build_library(verbose=False)
    assert path == LIBPATH and os.path.exists(path)
    lib = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/echopype_amd.h but not exported"
    # ... and the ctypes binding covers the same set
    from echopype_amd import _lib

    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert _lib.lib.epa_version() == 104


def test_library_carries_the_digest_of_the_sources_it_was_built_from(tmp_path):
    """epa_source_digest() == build.source_digest() of the tree, and the binding refuses a library whose digest differs
    (a stale shipped binary cannot answer for newer sources)."""
    from echopype_amd import _lib
    from echopype_amd.build import source_digest

    assert _lib.lib.epa_source_digest().decode() == source_digest()
    assert len(source_digest()) == 32
    # a source edit changes the digest the binding expects: the check raises
    import echopype_amd.build as b

    real = b.CSRC
    fake = tmp_path / "csrc"
    import shutil

    shutil.copytree(real, fake, ignore=shutil.ignore_patterns("_obj*"))
    with open(fake / "runtime.hip", "a") as f:
        f.write("// edited\n")
    b.CSRC = str(fake)
    try:
        assert b.source_digest() != _lib.lib.epa_source_digest().decode()
    finally:
        b.CSRC = real


def test_library_contains_gfx950_code_object_only():
    from echopype_amd.build import LIBPATH

    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", LIBPATH], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("llvm-readelf unavailable")
    blob = open(LIBPATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def test_errors_are_reported_without_a_gpu():
    """Argument validation happens before any HIP call; HIP failures come back as status codes."""
    from echopype_amd import _lib

    with pytest.raises(ValueError, match="NULL array argument"):
        _lib.call("epa_sv_power", None, None, 1, 1, 1, 0, 0, None, None, 1, None)
    with pytest.raises(ValueError, match="bad cal_type"):
        _lib.call("epa_sv_power", 8, 8, 1, 1, 1, 7, 0, 8, None, 1, None)
    n = ctypes.c_int()
    st = _lib.lib.epa_device_count(ctypes.byref(n))
    import torch

    if not torch.cuda.is_available():
        assert st == _lib.EPA_EHIP and "hipGetDeviceCount" in _lib.last_error()


def test_no_product_module_imports_the_oracle():
    """The oracle is test infrastructure: nothing under echopype_amd/ may import it."""
    pkg = os.path.join(ROOT, "echopype_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
