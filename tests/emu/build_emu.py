"""Build the CPU-emulated twin of libstgcn_hip.so (TEST INFRASTRUCTURE ONLY).

The unmodified product sources stgcn_amd/csrc/*.hip are compiled by the host clang++ against the
emulation shim tests/emu/hip/hip_runtime.h.  Returns the path of the shared library, rebuilding only
when a source is newer."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EMU = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "stgcn_amd", "csrc")
OUT = os.path.join(EMU, "_build", "libstgcn_emu.so")


def find_clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++") or ""):
        if c and os.path.exists(c):
            return c
    return None


def build(force=False):
    cxx = find_clang()
    if cxx is None:
        raise RuntimeError("host clang++ not found (needed for ext_vector_type)")
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(EMU, "emu_runtime.cpp"), os.path.join(EMU, "hip", "hip_runtime.h"),
            os.path.join(ROOT, "include", "stgcn_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [cxx, "-std=c++17", "-O2", "-fPIC", "-shared", "-x", "c++", "-I", EMU, "-DSTGCN_BACKEND_NAME=\"emu-cpu\"",
           "-Wno-unused-value", "-Wno-vla-cxx-extension",
           os.path.join(CSRC, "stgcn_capi.hip"), os.path.join(EMU, "emu_runtime.cpp"), "-o", OUT]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
