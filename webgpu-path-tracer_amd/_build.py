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


def build_lib(force=False, extra_flags=()):
    srcs = [os.path.join(CSRC, "ptmi.hip"), os.path.join(CSRC, "ptmi_bvh_device.hip"), os.path.join(CSRC, "ptmi_host.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("ptmi_device.h", "ptmi_kernels.h")] + [
        os.path.join(ROOT, "include", "ptmi.h"), os.path.join(ROOT, "include", "ptmi_math.h")]
    if not force and _newer(LIB, deps):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    _run([hipcc] + HIP_FLAGS + list(extra_flags) + ["-o", LIB] + srcs)
    return LIB


def build_variant(name, extra_flags):
    """A/B builds of the library with extra -D flags: webgpu-path-tracer_amd/variants/libptmi_<name>.so; run with PTMI_LIB=<path>."""
    vdir = os.path.join(PKG, "variants")
    os.makedirs(vdir, exist_ok=True)
    out = os.path.join(vdir, "libptmi_%s.so" % name)
    srcs = [os.path.join(CSRC, "ptmi.hip"), os.path.join(CSRC, "ptmi_bvh_device.hip"), os.path.join(CSRC, "ptmi_host.cpp")]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    _run([hipcc] + HIP_FLAGS + list(extra_flags) + ["-o", out] + srcs)
    return out


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
    print(build_addon(force=True))
    print(build_oracle(force=True))
