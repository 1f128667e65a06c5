"""Build recipes for the native pieces (explicit hipcc / gcc commands, outputs in-tree).

libptmi.so   = csrc/ptmi.hip (HIP kernels + C ABI, gfx950) + csrc/ptmi_bvh_device.hip (BVH build on the GPU, rocPRIM)
               + csrc/ptmi_host.cpp (host natives)
ptmi.node    = csrc/ptmi_napi.c (raw N-API binding of include/ptmi.h for the Node host), if the
               Node headers are present.
-ffp-contract=off and no fast-math are part of the numerical contract (include/ptmi_math.h).
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libptmi.so")
ADDON = os.path.join(PKG, "js", "ptmi.node")

HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    # no SLP vectoriser: on gfx950 a packed-f32 instruction issues in 4 cycles, the two scalar ones it replaces in 2 + 2, and the
    # packing costs moves and registers (k_bvh 92 -> 68 VGPRs, k_shade 92 -> 79, k_shade -6 %, k_generate -10 %; tools/kernel_resources.sh)
    "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function",
]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


SOURCES = ("ptmi.hip", "ptmi_bvh_device.hip", "ptmi_host.cpp")
TESTHOOKS_LIB = os.path.join(PKG, "variants", "libptmi_testhooks.so")


def _deps():
    return [os.path.join(CSRC, f) for f in SOURCES + ("ptmi_device.h", "ptmi_kernels.h")] + [os.path.join(ROOT, "include", "ptmi.h"), os.path.join(ROOT, "include", "ptmi_math.h")]


def _link(out, units):
    """units: [(source file name, extra flags, tag)].  One hipcc -c per translation unit, all at once (the three compile independently: ptmi.hip
    is ~45 s of the 55 a single command takes), then one link.  Objects: webgpu-path-tracer_amd/build/ (not shipped)."""
    from concurrent.futures import ThreadPoolExecutor

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    odir = os.path.join(PKG, "build")
    os.makedirs(odir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    flags = [f for f in HIP_FLAGS if f != "-shared"]
    objs = [os.path.join(odir, "%s%s.o" % (os.path.splitext(src)[0], tag)) for src, _, tag in units]
    with ThreadPoolExecutor(max_workers=len(units)) as ex:
        list(ex.map(lambda u: _run([hipcc] + flags + list(u[0][1]) + ["-c", os.path.join(CSRC, u[0][0]), "-o", u[1]]), zip(units, objs)))
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_lib(force=False, extra_flags=()):
    if not force and _newer(LIB, _deps()):
        return LIB
    return _link(LIB, [(src, tuple(extra_flags), "") for src in SOURCES])


def build_testhooks(force=False):
    """The tests' own build of the library: -DPTMI_TEST_HOOKS compiles in the fault injection the product build does not carry
    (PTMI_TEST_ALLOC_LIMIT: device allocations above N bytes fail; PTMI_TEST_RCCL_FAIL=init|reduce|mid: that RCCL call reports an error).
    Only ptmi.hip differs; the tests load it next to the product library (ptmi.load_library(path=...))."""
    if not force and _newer(TESTHOOKS_LIB, _deps()):
        return TESTHOOKS_LIB
    return _link(TESTHOOKS_LIB, [(src, ("-DPTMI_TEST_HOOKS",), "_testhooks") for src in SOURCES])


def build_variant(name, extra_flags):
    """A/B builds of the library with extra -D flags: webgpu-path-tracer_amd/variants/libptmi_<name>.so; run with PTMI_LIB=<path>."""
    return _link(os.path.join(PKG, "variants", "libptmi_%s.so" % name), [(src, tuple(extra_flags), "_" + name) for src in SOURCES])


def build_addon(force=False):
    src = os.path.join(CSRC, "ptmi_napi.c")
    inc = "/usr/include/node"
    if not os.path.exists(src) or not os.path.exists(os.path.join(inc, "node_api.h")):
        return None
    if not force and _newer(ADDON, [src, os.path.join(ROOT, "include", "ptmi.h")]):
        return ADDON
    _run(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-Wall", "-DNODE_GYP_MODULE_NAME=ptmi", "-I", inc, "-I", os.path.join(ROOT, "include"), "-o", ADDON, src, "-ldl"])
    return ADDON


def build_oracle(force=False):
    odir = os.path.join(ROOT, "oracle")
    out = os.path.join(odir, "libptm_oracle.so")
    if not force and _newer(out, [os.path.join(odir, "ptm_oracle.cpp"), os.path.join(ROOT, "include", "ptmi_math.h")]):
        return out
    _run(["make", "-C", odir, "-B"])
    return out


if __name__ == "__main__":
    print(build_lib(force=True))
    print(build_testhooks(force=True))
    print(build_addon(force=True))
    print(build_oracle(force=True))
