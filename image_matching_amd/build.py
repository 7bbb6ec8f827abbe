"""Compile libimatch.so for gfx950 with hipcc (cross-compiles without a GPU).

Every csrc/*.hip is one translation unit (no relocatable device code: no kernel calls across files), compiled to an
object under csrc/build/ -- in parallel, and only when the source or a header is newer -- and the objects are linked
into image_matching_amd/libimatch.so."""
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
OUT = os.path.join(HERE, "libimatch.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
# knn.hip: the MFMA kernels' hot epilogue is a max tree over accumulator registers; without this flag every fmaxf is
# preceded by a canonicalising v_max (5 vector instructions per 16x16 block instead of 3). No NaN can occur there
# (integer-valued products, -inf only as a pad row's start value); the one NaN-sensitive test of the file, the integrality
# check of uploaded rows, works on bit patterns.
EXTRA_FLAGS = {"knn.hip": ["-fno-honor-nans"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "imatch.h")]


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps)


def needs_build():
    return _stale(OUT, sources() + headers())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = headers()
    todo = [s for s in sources() if force or _stale(_obj(s), [s] + hdrs)]

    def compile_one(src):
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    jobs = max(1, min(len(todo), int(os.environ.get("MAX_JOBS", "8")), os.cpu_count() or 1))
    if todo:
        with ThreadPoolExecutor(jobs) as ex:
            list(ex.map(compile_one, todo))
    cmd = [hipcc] + FLAGS + ["-shared", "-o", OUT] + [_obj(s) for s in sources()]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    import sys
    build(force="--force" in sys.argv, verbose=True)
