"""Build libbocf_hip.so (gfx950) in-tree with hipcc.  `python -m bocf_amd.build [--force] [--asan-host | --probes]`.

Every translation unit is compiled to its own object (in parallel, only when it or a header changed) and the objects are
linked into bocf_amd/lib/libbocf_hip.so.  `--asan-host` builds the same sources with AddressSanitizer on the HOST side only
(-fsanitize=address -fno-gpu-sanitize: kernels are compiled as usual, uninstrumented) into bocf_amd/lib/libbocf_hip_asan.so:
the sanitizer target of the C-ABI shim's argument-validation paths (run on the CPU; GPU ASan is not available on this
pool)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libbocf_hip.so")
LIB_ASAN = os.path.join(LIBDIR, "libbocf_hip_asan.so")
LIB_PROBES = os.path.join(LIBDIR, "libbocf_hip_probes.so")
SOURCES = ["gemm_f64.hip", "gemm_f32.hip", "gemm_i8.hip", "fit.hip", "potrf.hip", "infer128.hip", "chol_team.hip", "hmc_stream.hip", "predict.hip", "acq.hip", "comm.hip", "capi.hip", "capi_chol.hip", "capi_fit.hip", "optimizer_host.hip"]
HEADERS = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + [os.path.join(os.path.dirname(HERE), "include", "bocf_hip.h")]


def hipcc_path():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    return _stale(LIB, [os.path.join(CSRC, f) for f in SOURCES] + HEADERS)


def _compile(src, obj, flags, verbose):
    cmd = [hipcc_path()] + flags + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)


def _build(lib, objdir, cflags, ldflags, force, verbose):
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + HEADERS):
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for f in [ex.submit(_compile, src, obj, cflags, verbose) for src, obj in jobs]:
            f.result()
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or not os.path.exists(lib):
        cmd = [hipcc_path()] + ldflags + objs + ["-o", lib, "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return lib


def build(force=False, verbose=True):
    """Compile every HIP translation unit for gfx950 and link the C-ABI shared library."""
    if not force and not needs_build():
        return LIB
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    return _build(LIB, OBJDIR, flags, ["--offload-arch=gfx950", "-shared", "-fPIC"], force, verbose)


def build_probes(force=False, verbose=False):
    """The same sources with -DBOCF_PROBES: the timing-only kernel variants (wrong results) and the test hooks (diagonal shift that
    forces the jitter ladder, single-process stand-in for the ranks of a sharded fit, forced schedule time-out / CU count) exist ONLY
    in this library; tools/ and the tests that need a hook load it (bocf_amd._ffi, env BOCF_PROBES=1), the product never does."""
    if not force and not _stale(LIB_PROBES, [os.path.join(CSRC, f) for f in SOURCES] + HEADERS):
        return LIB_PROBES
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DBOCF_PROBES"]
    return _build(LIB_PROBES, OBJDIR + "_probes", flags, ["--offload-arch=gfx950", "-shared", "-fPIC"], force, verbose)


def build_asan_host(force=False, verbose=False):
    """AddressSanitizer build of the host side of the same sources (device code uninstrumented: -fno-gpu-sanitize)."""
    if not force and not _stale(LIB_ASAN, [os.path.join(CSRC, f) for f in SOURCES] + HEADERS):
        return LIB_ASAN
    flags = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address", "-fno-gpu-sanitize", "-fno-omit-frame-pointer"]
    return _build(LIB_ASAN, OBJDIR + "_asan", flags, ["--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-fno-gpu-sanitize"],
                  force, verbose)


if __name__ == "__main__":
    if "--probes" in sys.argv:
        print(build_probes(force="--force" in sys.argv, verbose=True))
    elif "--asan-host" in sys.argv:
        print(build_asan_host(force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv))
