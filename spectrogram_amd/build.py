"""Builds libsonostft.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

  python -m spectrogram_amd.build          # incremental
  python -m spectrogram_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting spectrogram_amd/lib/libsonostft.so travels to the GPU box as is.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libsonostft.so")

ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# the tiled kernel's instantiations come first: they take longest (one translation unit per mode, compiled side by side)
HIP_SOURCES = ["sono_tile_u8.hip", "sono_tile_mag.hip", "sono_tile_band.hip", "sono_tile_survey.hip", "sono_tile.hip", "sono_kernels.hip", "sono_detect.hip",
               "sono_capi.hip", "sono_capi_detect.hip", "sono_capi_rois.hip", "sono_capi_noalloc.hip", "sono_stream.hip", "sono_measure_dev.hip"]
CXX_SOURCES = ["sono_host.cpp", "sono_framing.cpp", "sono_measure.cpp"]
# -fno-slp-vectorize: gfx950 executes v_pk_{add,mul,fma}_f32 at half the rate of the scalar forms
# (measured, tools/microbench.hip), so packing buys nothing and costs the moves that build the pairs.
HIP_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-slp-vectorize", "-Wall",
             "-Wno-unused-function"]
# The measurement stage reproduces double expression sequences of the reference: no fused multiply-adds, on the GPU
# as on the host (CXX_FLAGS below).
HIP_EXTRA = {"sono_measure_dev.hip": ["-ffp-contract=off"]}
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wextra"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _all_headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "sonostft.h"))
    return hs


# --asan: the same library with AddressSanitizer and UndefinedBehaviorSanitizer on the HOST code of every
# translation unit (the C ABI, the framing and measurement stages; device code is compiled as always: there is
# no GPU sanitizer on this hardware pool).  Loaded through SONO_LIB_PATH by tools/asan_tests.sh.
ASAN_LIB = os.path.join(LIBDIR, "libsonostft_asan.so")
ASAN_OBJDIR = os.path.join(HERE, "build", "asan")
# (-fno-sanitize=function: a kernel launched through a pointer to its host stub reads as a call through a wrong function type)
ASAN_HOST = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host", "-fno-sanitize=function",
             "-Xarch_host", "-fno-omit-frame-pointer", "-g"]


def build(force=False, verbose=False, asan=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir, lib_path = (ASAN_OBJDIR, ASAN_LIB) if asan else (OBJDIR, LIB)
    os.makedirs(objdir, exist_ok=True)
    headers = _all_headers()
    objs = []
    jobs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([HIPCC] + HIP_FLAGS + HIP_EXTRA.get(src, []) + (ASAN_HOST if asan else []) + ["-c", s, "-o", o])
    for src in CXX_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            if asan:        # one sanitizer runtime for the whole library: clang's, through hipcc
                jobs.append([HIPCC, "-x", "c++"] + CXX_FLAGS + ["-fsanitize=address", "-fsanitize=undefined", "-fno-omit-frame-pointer", "-g",
                                                              "-c", s, "-o", o])
            else:
                jobs.append(["g++"] + CXX_FLAGS + ["-c", s, "-o", o])
    procs = []
    for cmd in jobs:
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("build failed: " + " ".join(cmd))
    if jobs or force or not os.path.exists(lib_path):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib_path] + objs
        if asan:
            cmd += ["-fsanitize=address", "-fsanitize=undefined", "-shared-libsan"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, asan="--asan" in sys.argv))
