"""The C-ABI library loads and exports every symbol include/ptmi.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.load_library()
    hdr = open(os.path.join(ROOT, "include", "ptmi.h")).read()
    declared = sorted(set(re.findall(r"\b(ptmi_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no prototypes parsed"
    assert sorted(pkg.ptmi.SYMBOLS) == declared
    for name in declared:
        assert hasattr(L, name), name
    assert L.ptmi_version() == 5


def test_the_tests_own_build_of_the_library_is_the_same_abi(pkg, hooks):
    """webgpu-path-tracer_amd/variants/libptmi_testhooks.so (-DPTMI_TEST_HOOKS: the fault injection tests/ use, compiled out of the product library)
    loads next to the product library and exports the same surface; the product library's text does not even contain the hooks' variable names."""
    for name in pkg.ptmi.SYMBOLS:
        assert hasattr(hooks, name), name
    assert hooks.ptmi_version() == pkg.load_library().ptmi_version()
    assert b"PTMI_TEST_" in open(pkg._build.TESTHOOKS_LIB, "rb").read()
    assert b"PTMI_TEST_" not in open(pkg.ptmi.lib_path(), "rb").read()


def test_status_strings_and_defaults(pkg):
    L = pkg.load_library()
    assert L.ptmi_status_string(0) == b"ok"
    assert b"scene" in L.ptmi_status_string(-5)
    p = pkg.default_params()
    # shaders/header.wgsl:9-13, traceRay.wgsl:8, main.wgsl:7
    assert (p.num_samples, p.max_bounces, p.stratify, p.importance_sampling, p.stack_size) == (1, 100, 0, 0, 20)
    assert list(p.background) == [0.0, 1.0, 1.0] and p.fov_degrees == 60.0


def test_struct_sizes_match_header(pkg):
    assert ctypes.sizeof(pkg.Params) == 4 * 5 + 12 + 4 + 4 + 20
    assert ctypes.sizeof(pkg.ptmi.Stats) == 12 * 8 + 8 * 8 + 3 * 8 + 2 * 8 + 4 * 8  # (+ tail_ms, tail_launches: k_tail; API v4: reduce_mode, peer_links, placement_*)
    assert pkg.ptmi.HIT_DTYPE.itemsize == 4 + 4 + 12 + 12 + 4 + 64


def test_create_without_gpu_fails_loudly(pkg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.PtmiError) as e:
        pkg.Context(0)
    assert e.value.status == -2 and "HIP" in str(e.value)


def test_null_context_calls_return_invalid_arg(pkg):
    L = pkg.load_library()
    assert L.ptmi_upload(None, 1, None, 0) == -1
    assert L.ptmi_resize(None, 4, 4) == -1
    assert L.ptmi_synchronize(None) == -1
    L.ptmi_destroy(None)
