"""Build libsage_hip.so (HIP kernels + C ABI + host index builder) for gfx950, in-tree.

    python -m sage_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off on EVERY translation unit: the reference's f32
arithmetic (Tolerance::bounds, ion masses, ppm sums) decides matches, and hipcc/clang would otherwise
contract a*b+c into an FMA.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("SAGE_HIP_LIB") or os.path.join(HERE, "libsage_hip.so")  # (override: kernel experiments)
SOURCES = ["kernels.hip", "process.hip", "index_build.hip", "rescore.hip", "capi.hip", "host_db.cpp", "writers.cpp", "mzml_reader.cpp"]
HEADERS = ["core.h", "crlog.h", "crlog_tables.h", "detmath.h", "device_types.h", "host_db.hpp", os.path.join("..", "..", "include", "sage_hip.h")]
ARCH = "gfx950"


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _flags():
    common = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
    return common + os.environ.get("SAGE_HIP_EXTRA_FLAGS", "").split()


def source_digest():
    """sha256 over every source and header of the library (names + contents), this file and the compiler flags: what the
    shipped .so was built FROM.  Modification times say nothing on a fresh checkout or on the GPU box's snapshot."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(set([os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)])):
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(" ".join([ARCH] + _flags()).encode())
    return h.hexdigest()


def _stamp():
    return LIB + ".srchash"


def needs_build():
    """True unless the library exists AND carries the digest of the sources / flags in the tree now."""
    if not os.path.exists(LIB) or not os.path.exists(_stamp()):
        return True
    with open(_stamp()) as fh:
        return fh.read().strip() != source_digest()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    common = _flags()
    digest = source_digest()
    # objects of the default build beside the sources; a variant's (SAGE_HIP_OBJ_SUFFIX, scripts/variants.sh) under build/,
    # which neither git nor the GPU box's snapshot carries
    suffix = os.environ.get("SAGE_HIP_OBJ_SUFFIX", "")
    objdir = CSRC if not suffix else os.path.join(HERE, "..", "build", "variants", suffix.strip("_"))
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc(), f"--offload-arch={ARCH}", *common, "-c", os.path.join(CSRC, src), "-o", obj]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "hip")
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:  # translation units are independent
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB, "-lpthread", "-lz"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(_stamp(), "w") as fh:
        fh.write(digest + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
