"""Build libria_gpu.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libria_gpu.so")
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
               "-fno-slp-vectorize",   # SLP packs adjacent f32 adds/muls into v_pk_*_f32 + register shuffles: measured slower on gfx950
               "-fPIC", "-shared", "-std=c++17", "-Wno-inline-asm", "-Wno-pass-failed"]


def _sources():
    out = [os.path.join(HERE, "..", "include", "ria_gpu.h")]
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h", ".hpp", ".inc")):
            out.append(os.path.join(CSRC, f))
    return out


STAMP = LIB + ".flags"   # the compiler flags the library on disk was built with (one line)


def flags_line():
    return " ".join(HIPCC_FLAGS)


def needs_build():
    """A library is current when it is newer than every source AND was built with today's flags (a prebuilt library
    shipped without its stamp counts as current only where there is no compiler to rebuild it: see build())."""
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(s) > t for s in _sources()):
        return True
    try:
        return open(STAMP).read().strip() != flags_line()
    except OSError:
        return True


def under_profiler():
    """rocprofv3 (and friends) preload a tool library into every child: a hipcc started from here would inherit it, which
    this pool forbids (the profiler initialises the GPU in each process it is loaded into)."""
    env = os.environ
    return any("rocprof" in env.get(k, "").lower() or "roctracer" in env.get(k, "").lower() for k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES")) \
        or "ROCPROFILER_LIBRARY_PATH" in env


def build(force=False, verbose=False):
    """Compile ria_amd/csrc/ria_gpu.hip -> ria_amd/lib/libria_gpu.so. Returns the library path.

    Safe when several processes of one job (torchrun ranks) get here at once: one compiles under an exclusive file
    lock, into a temporary name that is renamed into place; the others wait, re-check and load the finished library."""
    if not force and not needs_build():
        return LIB
    if under_profiler():
        if os.path.exists(LIB) and not force:
            raise RuntimeError("libria_gpu.so is out of date and this process runs under a profiler: build first (python -m ria_amd.build), "
                               "then profile - the compiler must not be started from a profiled process")
        raise RuntimeError("refusing to start hipcc from a profiled process: run python -m ria_amd.build first")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(LIB):
            return LIB  # prebuilt library shipped with the snapshot, no compiler on this box
        raise RuntimeError("hipcc not found and no prebuilt libria_gpu.so")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    import fcntl
    with open(os.path.join(os.path.dirname(LIB), ".build.lock"), "a") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB      # another process built it while this one waited
            tmp = f"{LIB}.tmp.{os.getpid()}"
            cmd = [hipcc] + HIPCC_FLAGS + ["-o", tmp, os.path.join(CSRC, "ria_gpu.hip")]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd, cwd=CSRC)
                os.replace(tmp, LIB)
                with open(STAMP, "w") as f:
                    f.write(flags_line() + "\n")
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
