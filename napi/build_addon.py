"""Builds the Node N-API addon napi/sonostft.node (plain C, gcc) against the
in-tree libsonostft.so.  Skipped when the Node headers are not installed."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "sonostft.node")
NODE_INCLUDE = "/usr/include/node"


def build(force=False):
    src = os.path.join(HERE, "sonostft_napi.c")
    libdir = os.path.join(ROOT, "spectrogram_amd", "lib")
    if not os.path.exists(os.path.join(NODE_INCLUDE, "node_api.h")):
        return None
    deps = [src, os.path.join(ROOT, "include", "sonostft.h"), os.path.join(libdir, "libsonostft.so")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c99", "-Wall", "-Wextra", "-DNODE_GYP_MODULE_NAME=sonostft",
                           "-DNAPI_VERSION=4", "-I", NODE_INCLUDE, src, "-o", OUT, "-L", libdir, "-lsonostft",
                           "-Wl,-rpath,$ORIGIN/../spectrogram_amd/lib"])
    return OUT


if __name__ == "__main__":
    print(build(force=True))
