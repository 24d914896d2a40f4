"""Builds libmoonshine_amd.so (HIP kernels + C ABI + host-side scene I/O) for gfx950 with hipcc, in-tree, and the
`offline` CLI that links against it.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract of the
hot path (every f32 operation is a single IEEE operation, see csrc/msne_math.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
SOURCES = ["context.hip", "bvh_build.hip", "trace.hip", "integrator.hip", "env.hip", "group.hip"]
HOST_SOURCES = ["exr.cpp", "png.cpp", "glb.cpp", "scene_io.cpp"]      # plain C++ above the C ABI
LIB = os.path.join(HERE, "libmoonshine_amd.so")
OFFLINE = os.path.join(HERE, "offline")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")
EXTRA = os.environ.get("MSNE_CXXFLAGS", "").split()
# -fno-slp-vectorize: the SLP vectoriser packs adjacent scalar f32 multiplies / adds into v_pk_mul_f32 / v_pk_add_f32, which issue in 4 cycles for two
# operations — no faster than two 2-cycle scalar ones (profiles/r03_valu_microbench.txt) — and cost moves and registers to pair their operands:
# without it S1 +1.3 %, S2 +5.0 %, and k_shade capped at 128 registers spills 35 instead of 65 (profiles/r03_compiler_flags.txt).  Same IEEE operations either way.
FLAGS = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-Wno-comment"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-comment"]


# per-source flags.  trace.hip: the scheduler's max-ILP strategy (S2's closest-hit kernel -2.6 %, S1 and the sky scene within noise; iterative-minreg: -23 %)
SOURCE_FLAGS = {"trace.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
for _src in list(SOURCE_FLAGS) + ["integrator.hip"]:   # experiments: $MSNE_FLAGS_trace_hip="..." replaces a source's own flags
    if os.environ.get("MSNE_FLAGS_" + _src.replace(".", "_")) is not None:
        SOURCE_FLAGS[_src] = os.environ["MSNE_FLAGS_" + _src.replace(".", "_")].split()


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force=False, verbose=False, variant=None, extra_flags=()):
    """variant: build a second library libmoonshine_amd_<variant>.so with extra compiler flags (tests use it to run the
    traversal with a one-entry LDS stack, i.e. through the HBM spill path); the default build is untouched."""
    lib = LIB if variant is None else os.path.join(HERE, "libmoonshine_amd_%s.so" % variant)
    flags = list(extra_flags) + FLAGS
    api_h = os.path.join(HERE, "..", "include", "moonshine_amd.h")
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [api_h]
    objdir = os.path.join(HERE, "build" if variant is None else "build_" + variant)
    os.makedirs(objdir, exist_ok=True)
    # objects built with other flags are stale too
    stamp, want = os.path.join(objdir, "flags.txt"), " ".join(flags + HOST_FLAGS + [repr(sorted(SOURCE_FLAGS.items()))])
    if not os.path.exists(stamp) or open(stamp).read() != want:
        force = True
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + flags + SOURCE_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src in HOST_SOURCES:
        s = os.path.join(HOST, src)
        o = os.path.join(objdir, "host_" + src.replace(".cpp", ".o"))
        objs.append(o)
        if force or _stale(o, [s, os.path.join(HOST, "host.h"), api_h]):
            cmd = [CXX] + HOST_FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed on %s" % src)
    if force or procs or _stale(lib, objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lz", "-ldl", "-lpthread"])
    open(stamp, "w").write(want)
    if variant is not None:
        return lib
    cli = os.path.join(HOST, "offline.cpp")
    if force or _stale(OFFLINE, [cli, LIB, api_h]):
        subprocess.check_call([CXX] + HOST_FLAGS + ["-o", OFFLINE, cli, "-L" + HERE, "-lmoonshine_amd", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
