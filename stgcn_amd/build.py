"""Build libstgcn_hip.so for gfx950 with hipcc (in-tree, next to this file).

    python -m stgcn_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the repo snapshot."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libstgcn_hip.so")
SOURCES = ["stgcn_capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSTGCN_BACKEND_NAME=\"hip-gfx950\""]


def find_hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build stgcn_amd)")


def needs_build():
    if not os.path.exists(OUT):
        return True
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "stgcn_hip.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [find_hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
