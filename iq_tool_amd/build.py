"""Builds libiqgpu.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m iq_tool_amd.build

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libiqgpu.so")
SOURCES = ["design.cpp", "abi.cpp", "plan.cpp", "process.cpp", "agc_host.cpp", "pipeline.cpp", "iq_optimizer.cpp", "wav_meta.cpp", "topology.cpp", "kernels.hip", "front_wave.hip", "front_fat.hip", "front_mid.hip", "front_p0.hip", "front_s2.hip", "cascade_wave.hip", "cascade2.hip", "fftconv.hip", "p0fft_cu8.hip", "p0fft_cs8.hip", "p0fft_cs16.hip", "interp.hip", "agc.hip"]
HEADERS = ["design.hpp", "chain.hpp", "kernels.hpp", "dsp_device.hpp", "wave_common.hpp", "front_tiles.hpp", "cascade_tiles.hpp", "front_fat_common.hpp", "front_p0_common.hpp", "fft16.hpp", "p0fft.hpp", os.path.join("..", "..", "include", "iqgpu.h")]
# headers only some sources include: {header: prefixes of the sources that depend on it} (the rest depend on every header)
PRIVATE_HEADERS = {"fft16.hpp": ("fftconv.hip", "p0fft_"), "p0fft.hpp": ("p0fft_",)}
HARNESS_SRC = os.path.join(CSRC, "harness", "iqgpu_run.c")
HARNESS_BIN = os.path.join(LIBDIR, "iqgpu_run")


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libiqgpu cannot be built (no CPU fallback exists)")


def _stale(target, deps):
    deps = list(deps) + [os.path.abspath(__file__)]     # (flags live in this file)
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-Wall"]
# per-source flags.  front_mid.hip: the tile claim is ONE lane's returning atomic add whose value is read a tile later; the atomic
# optimizer would rewrite it as a wave reduction + broadcast and wait for the result on the spot (tools/check_isa.py guards it)
SOURCE_FLAGS = {"front_mid.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}


def build_lib(force=False, verbose=False, extra_flags=(), out=None):
    """one object per source (compiled in parallel, rebuilt only when the source or a header is newer), then the link"""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    lib = out or LIB
    import hashlib
    # (a stable digest: str hashes are randomised per process, which used to defeat the incremental rebuild and pile up directories)
    objdir = os.path.join(LIBDIR, "obj" + ("" if not extra_flags else "_" + hashlib.sha256(" ".join(extra_flags).encode()).hexdigest()[:10]))
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for sname in SOURCES:
        src, obj = os.path.join(CSRC, sname), os.path.join(objdir, sname + ".o")
        hdrs = [os.path.join(CSRC, h) for h in HEADERS if h not in PRIVATE_HEADERS or sname.startswith(PRIVATE_HEADERS[h])]
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc(), *FLAGS, *SOURCE_FLAGS.get(sname, []), *extra_flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=min(6, max(1, (os.cpu_count() or 2) - 1))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, sname + ".o") for sname in SOURCES]
    if jobs or force or _stale(lib, objs):
        run([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", lib])
    if os.path.exists(HARNESS_SRC) and (force or _stale(HARNESS_BIN, [HARNESS_SRC, LIB])):
        cmd = ["gcc", "-O2", "-std=gnu99", "-Wall", "-I", os.path.join(HERE, "..", "include"),
               HARNESS_SRC, "-o", HARNESS_BIN, "-L", LIBDIR, "-liqgpu", "-Wl,-rpath,$ORIGIN", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
