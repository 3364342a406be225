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
OUT_ASAN = os.path.join(EMU, "_build", "libstgcn_emu_asan.so")
OUT_RACE = os.path.join(EMU, "_build", "libstgcn_emu_race.so")


def find_clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++") or ""):
        if c and os.path.exists(c):
            return c
    return None


def asan_runtime():
    """The shared AddressSanitizer runtime of the host clang: a python process that loads the --asan build needs it LD_PRELOADed."""
    cxx = find_clang()
    if cxx is None:
        return None
    r = subprocess.run([cxx, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def build(force=False, asan=False, race=False):
    """asan=True: the same sources under -fsanitize=address (the emulator is the one place a sanitizer can look at these kernels: every
    global / LDS access of a kernel is an ordinary host access there).  Run with tools/emu_asan.sh.
    race=True: the LDS race check -- the sources compiled with -fsanitize=thread, whose __tsan_read / __tsan_write hooks are the emulator's OWN
    (emu_runtime.cpp: no ThreadSanitizer runtime is linked): every LDS access is checked for a conflicting access of another wave with no
    workgroup barrier in between.  STGCN_EMU_RACE=1 makes tests/emu_util.py bind this build (tools/emu_race.sh)."""
    global OUT
    if race:
        saved, OUT = OUT, OUT_RACE
        try:
            return _build(force, ["-fsanitize=thread", "-O1", "-g", "-DSTGCN_EMU_RACE=1"])
        finally:
            OUT = saved
    if asan:
        saved, OUT = OUT, OUT_ASAN
        try:
            return _build(force, ["-fsanitize=address", "-shared-libasan", "-fno-omit-frame-pointer", "-g", "-O1", "-DSTGCN_EMU_ASAN=1"])
        finally:
            OUT = saved
    return _build(force, ["-O2"])


def _build(force, opt):
    cxx = find_clang()
    if cxx is None:
        raise RuntimeError("host clang++ not found (needed for ext_vector_type)")
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(EMU, "emu_runtime.cpp"), os.path.join(EMU, "hip", "hip_runtime.h"),
            os.path.join(ROOT, "include", "stgcn_hip.h")]
    def fresh():
        return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs)

    if not force and fresh():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # Several processes may get here at once (the spawned ranks of tests/test_dp_gloo.py on a fresh checkout): one builds, the others wait
    # for it; the library appears by an atomic rename, so nobody ever opens a half-written file.
    import fcntl
    with open(os.path.join(os.path.dirname(OUT), ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or not fresh():
                tmp = OUT + ".tmp.%d" % os.getpid()
                cmd = [cxx, "-std=c++17", *opt, "-fPIC", "-shared", "-x", "c++", "-I", EMU, "-DSTGCN_BACKEND_NAME=\"emu-cpu\"",
                       "-Wno-unused-value", "-Wno-vla-cxx-extension",
                       os.path.join(CSRC, "stgcn_capi.hip"), os.path.join(EMU, "emu_runtime.cpp"), "-o", tmp]
                try:
                    subprocess.run(cmd, check=True)
                    os.replace(tmp, OUT)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return OUT


if __name__ == "__main__":
    import sys
    print(build(force=True, asan="--asan" in sys.argv, race="--race" in sys.argv))
