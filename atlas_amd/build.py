"""Build the gfx950 shared library (and the host-only helper library used by CPU tests).

    python -m atlas_amd.build          # builds atlas_amd/lib/libatlas_hip.so (+ libatlas_host.so)

hipcc cross-compiles for gfx950 without a GPU. The .so files are git-ignored but travel with
the repo snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
HIP_SO = os.path.join(LIBDIR, "libatlas_hip.so")
TUNE_SO = os.path.join(LIBDIR, "libatlas_hip_tune.so")     # -DATLAS_TUNING=1: knobs + stamps for tools/ and the configuration tests
HOST_SO = os.path.join(LIBDIR, "libatlas_host.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def build_hip(force=False, verbose=False, tuning=False):
    """the product library (no knobs, no mutable globals) or, with tuning=True, the tuning build of the same sources"""
    os.makedirs(LIBDIR, exist_ok=True)
    HIP_SO = TUNE_SO if tuning else globals()["HIP_SO"]
    srcs = [os.path.join(CSRC, "atlas_hip.hip"), os.path.join(CSRC, "common.h"),
            os.path.join(HERE, "..", "include", "atlas_hip.h")]
    extra = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    if force or _newer(HIP_SO, srcs + extra):
        hip_srcs = sorted(f for f in extra if f.endswith(".hip"))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
               "-Wall", "-Wno-unused-function",
               # MFMA results stay in architectural VGPRs: the epilogues / softmax work on them with VALU, and hipcc's default
               # AGPR form paid one v_accvgpr_read per element (6 700 of them across encoder.hip)
               "-mllvm", "-amdgpu-mfma-vgpr-form=1", *(["-DATLAS_TUNING=1"] if tuning else []), *hip_srcs, "-o", HIP_SO]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
        except subprocess.CalledProcessError:
            # a hipcc without the MFMA-form switch: same code, AGPR-form MFMAs (slower epilogues / softmax, same results)
            cmd = [c for c in cmd if c not in ("-mllvm", "-amdgpu-mfma-vgpr-form=1")]
            print("retrying without -amdgpu-mfma-vgpr-form:", " ".join(cmd))
            subprocess.check_call(cmd)
    return HIP_SO


def build_host(force=False, verbose=False):
    """common.h compiled with g++ for CPU-side unit tests of the device arithmetic helpers."""
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, "host_helpers.cpp"), os.path.join(CSRC, "common.h")]
    if force or _newer(HOST_SO, srcs):
        cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", srcs[0], "-o", HOST_SO]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_SO


def build_all(force=False, verbose=False):
    """product + tuning builds (side by side: two hipcc processes) + the host helper library"""
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(2) as ex:
        a = ex.submit(build_hip, force, verbose)
        b = ex.submit(build_hip, force, verbose, True)
        hip = a.result()
        b.result()
    return hip, build_host(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
