"""Build the native libraries in-tree.

  librt_mi355x.so  HIP kernels + the C ABI of include/rt_mi355x.h   (hipcc, gfx950)
  librt_host.so    C++ mirror of the reference's host-side scene API  (g++, CPU only)

Both are built with -ffp-contract=off: the canonical arithmetic forbids FMA contraction
(DESIGN.md).  hipcc keeps fp32 / and sqrt correctly rounded and keeps denormals by default;
the flags are passed explicitly so that a changed default cannot silently break id parity.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
DEVICE_LIB = os.path.join(HERE, "librt_mi355x.so")
HOST_LIB = os.path.join(HERE, "librt_host.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
HIP_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-fno-fast-math",
             "-Wall", "-Wno-bitwise-instead-of-logical"]
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-bitwise-instead-of-logical"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(*dirs):
    out = []
    for d in dirs:
        for f in os.listdir(d):
            if f.endswith((".h", ".hip", ".cpp")):
                out.append(os.path.join(d, f))
    return out


def build_device(force=False, extra_flags=()):
    srcs = _sources(CSRC) + [os.path.join(HERE, "..", "include", "rt_mi355x.h")]
    if force or _stale(DEVICE_LIB, srcs):
        cmd = [HIPCC] + HIP_FLAGS + list(extra_flags) + [os.path.join(CSRC, "rt_mi355x.hip"), "-o", DEVICE_LIB]
        subprocess.check_call(cmd)
    return DEVICE_LIB


def build_host(force=False):
    srcs = _sources(os.path.join(CSRC, "host"))
    if force or _stale(HOST_LIB, srcs):
        cmd = ["g++"] + CXX_FLAGS + [os.path.join(CSRC, "host", "host_capi.cpp"), "-o", HOST_LIB]
        subprocess.check_call(cmd)
    return HOST_LIB


def build_all(force=False):
    build_host(force)
    build_device(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built", DEVICE_LIB, HOST_LIB)
