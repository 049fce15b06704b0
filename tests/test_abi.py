"""The C-ABI library loads and exports every symbol include/dhqr.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dhqr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dhqr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(pkg):
    pkg.build()
    L = ctypes.CDLL(pkg._lib.SO_PATH)
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert hasattr(L, n), f"libdhqr.so does not export {n}"
    # the Python binding covers the whole header and nothing else
    assert sorted(pkg._lib.SIGNATURES) == names


def test_micro_benchmarks_are_not_in_the_product_library(pkg):
    """include/dhqr_bench.h is exported by libdhqr_bench.so (a superset build) and by it alone"""
    pkg.build()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dhqr_bench.h")).read(), flags=re.S)
    bench = sorted(set(re.findall(r"\b(dhqr_[a-z0-9_]+)\s*\(", src)))
    assert sorted(pkg._lib.BENCH_SIGNATURES) == bench and len(bench) >= 7
    L, B = ctypes.CDLL(pkg._lib.SO_PATH), ctypes.CDLL(pkg._lib.BENCH_SO_PATH)
    import subprocess
    exported = subprocess.run(["nm", "-D", "--defined-only", pkg._lib.SO_PATH], capture_output=True, text=True).stdout
    assert "dhqr_bench_" not in exported and "dhqr_debug_" not in exported
    for n in bench + _declared():
        assert hasattr(B, n), f"libdhqr_bench.so does not export {n}"
    for n in bench:
        assert not hasattr(L, n)


def test_layout_constants_match_the_header(pkg):
    hdr = open(os.path.join(ROOT, "include", "dhqr.h")).read()
    macro = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(DHQR_NB|DHQR_CS_BLOCK|DHQR_ZNB)\s+(\d+)", hdr)}
    assert macro["DHQR_NB"] == pkg._lib.NB and macro["DHQR_CS_BLOCK"] == pkg._lib.CS_BLOCK == 2 * pkg._lib.NB
    assert macro.get("DHQR_ZNB", pkg._lib.ZNB) == pkg._lib.ZNB


def test_version_and_error_string(pkg):
    L = pkg._lib.lib()
    assert L.dhqr_version() == 300
    assert isinstance(L.dhqr_last_error(), bytes)
    assert L.dhqr_panel_ldv(100) == 112 and L.dhqr_panel_ldv(128) == 128
    assert L.dhqr_panel_buffer_elems(256) == 256 * 128 + 2 * 128 * 128 + 128 + 16  # V | T | Tt | alpha | status


def test_no_cpu_fallback(pkg):
    """Without a GPU the product path must fail loudly (DHQR_ENODEVICE), never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.DHQRError) as e:
        pkg.Context(0)
    assert e.value.code == pkg._lib.ENODEVICE
    import numpy as np
    with pytest.raises(pkg.DHQRError):
        pkg.qr_(np.asfortranarray(np.random.rand(8, 4)))


def test_product_package_does_not_import_oracle():
    pkgdir = os.path.join(ROOT, "distributedhouseholderqr.jl_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert not re.search(r"#\s*include[^\n]*oracle", txt), f
                assert "libdhqr_oracle" not in txt and "dhqr_oracle_" not in txt.replace(
                    "dhqr_oracle_u01", ""), f  # (a comment names the generator twin)


def test_julia_wrapper_binds_existing_symbols_with_matching_arity():
    """The Julia `ccall` stubs (unexecutable here: no Julia) must at least name symbols the header
    declares, with as many argument types as the C prototype has parameters."""
    jl = open(os.path.join(ROOT, "distributedhouseholderqr.jl_amd", "julia", "DistributedHouseholderQR.jl")).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dhqr.h")).read(), flags=re.S)
    protos = {m.group(1): m.group(2) for m in re.finditer(r"\b(dhqr_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S)}
    calls = []
    for m in re.finditer(r"ccall\(\(:(dhqr_[a-z0-9_]+),\s*libdhqr\),\s*\w+,\s*\(", jl):
        i, depth = m.end(), 1          # balanced scan of the argument-type tuple
        while depth:
            depth += {"(": 1, ")": -1}.get(jl[i], 0)
            i += 1
        calls.append((m.group(1), jl[m.end(): i - 1]))
    assert len(calls) >= 5
    for name, argtypes in calls:
        assert name in protos, f"{name} is not declared in include/dhqr.h"
        nparams = 0 if protos[name].strip() in ("", "void") else protos[name].count(",") + 1
        flat = re.sub(r"\{[^{}]*\}", "", argtypes)   # Ptr{Cvoid}, Ref{Ptr{Cvoid}} -> no inner commas
        flat = re.sub(r"\{[^{}]*\}", "", flat)
        jl_args = [a for a in flat.replace("\n", " ").split(",") if a.strip()]
        assert len(jl_args) == nparams, (name, len(jl_args), nparams)
