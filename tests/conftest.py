import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_pkg():
    """The package directory has a hyphen: load it under the module name `webgpu_path_tracer_amd`."""
    name = "webgpu_path_tracer_amd"
    if name in sys.modules:
        return sys.modules[name]
    pdir = os.path.join(ROOT, "webgpu-path-tracer_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pdir, "__init__.py"), submodule_search_locations=[pdir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def oracle():
    from oracle import ptm_oracle

    ptm_oracle.build()
    return ptm_oracle


@pytest.fixture(scope="session")
def hooks(pkg):
    """The tests' own build of the library (-DPTMI_TEST_HOOKS: PTMI_TEST_ALLOC_LIMIT, PTMI_TEST_RCCL_FAIL — fault injection the product build does not
    carry), loaded NEXT TO the product library: Context(..., lib=hooks).  Built by __graft_entry__.build()."""
    return pkg.ptmi.load_library(path=pkg._build.TESTHOOKS_LIB)


@pytest.fixture(scope="session")
def ctx(pkg):
    """One device context for the whole GPU session.  No skip: on a GPU box a missing library or
    device must FAIL the gpu-marked tests."""
    c = pkg.Context(0)
    yield c
    c.close()


def canon_bits(a):
    """uint32 view with every NaN mapped to one pattern (x86 and gfx950 generate different default
    NaN signs; NaN-ness is what the shader semantics carry)."""
    a = np.ascontiguousarray(a, np.float32)
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7FC00000
    return b


def assert_same_bits(got, want, what=""):
    g, w = canon_bits(got), canon_bits(want)
    if g.shape != w.shape:
        raise AssertionError("%s: shape %s vs %s" % (what, g.shape, w.shape))
    bad = np.nonzero(g.reshape(-1) != w.reshape(-1))[0]
    if bad.size:
        gf, wf = np.asarray(got, np.float32).reshape(-1), np.asarray(want, np.float32).reshape(-1)
        i = bad[:8]
        raise AssertionError("%s: %d of %d values differ; first at %s: got %s want %s" % (what, bad.size, g.size, i, gf[i], wf[i]))


def cornell_view(pkg, name="cornell"):
    return pkg.scenes.camera_view(*pkg.scenes.CAMERAS[name])
