"""Compile libimatch.so for gfx950 with hipcc (cross-compiles without a GPU).

Every csrc/*.hip is one translation unit (no relocatable device code: no kernel calls across files), compiled to an
object under csrc/build/ -- in parallel, and only when its source, a header, its command line or the compiler has changed
-- and the objects are linked into image_matching_amd/libimatch.so.

* Staleness is decided by a stamp file beside every object and beside the library: a hash of the full command line, the
  compiler's version string and the contents of the source and of every header. Changing FLAGS / EXTRA_FLAGS, or
  overwriting the library from a development script, therefore rebuilds (modification times alone did not).
* Objects and the library are written under a temporary name and renamed into place, under an exclusive lock on
  csrc/build/.lock: several ranks importing the package at once compile once, and nobody loads a half-written file.
* IMATCH_LIB_OUT selects another output path for the library (development builds with extra -D flags,
  IMATCH_BUILD_DEFINES="-DKNN_TRACE ..."): the in-tree libimatch.so is then left alone.
* After knn.hip is compiled, its gfx950 machine code is audited for the MFMA SrcC write-after-read pattern
  (mfma_audit.py); a finding fails the build."""
import fcntl
import glob
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
OUT = os.environ.get("IMATCH_LIB_OUT") or os.path.join(HERE, "libimatch.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
# knn.hip: the MFMA kernels' hot epilogue is a max tree over accumulator registers; without this flag every fmaxf is
# preceded by a canonicalising v_max (5 vector instructions per 16x16 block instead of 3). No NaN can occur there
# (integer-valued products, -inf only as a pad row's start value); the one NaN-sensitive test of the file, the integrality
# check of uploaded rows, works on bit patterns.
EXTRA_FLAGS = {"knn.hip": ["-fno-honor-nans"]}
AUDITED = ("knn.hip",)   # translation units whose machine code goes through mfma_audit


def hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "imatch.h")]


def _defines():
    return os.environ.get("IMATCH_BUILD_DEFINES", "").split()


def _obj(src):
    tag = ""
    if _defines():   # development objects do not overwrite the product's
        tag = "." + hashlib.sha1(" ".join(_defines()).encode()).hexdigest()[:8]
    return os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0] + tag + ".o")


_version = None


def _compiler_version():
    global _version
    if _version is None:
        try:
            _version = subprocess.check_output([hipcc(), "--version"], stderr=subprocess.STDOUT).decode()
        except (OSError, subprocess.CalledProcessError) as e:
            _version = "unknown: %s" % e
    return _version


def _compile_cmd(src):
    return [hipcc()] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + _defines() + ["-c", src, "-o", _obj(src)]


def _link_cmd():
    return [hipcc()] + FLAGS + ["-shared", "-o", OUT] + [_obj(s) for s in sources()]


def _digest(cmd, files):
    h = hashlib.sha256()
    h.update("\0".join(cmd).encode())
    h.update(_compiler_version().encode())
    for p in files:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_of(target):
    try:
        with open(target + ".stamp") as f:
            return f.read().strip()
    except OSError:
        return None


def _obj_digest(src):
    return _digest(_compile_cmd(src), [src] + headers())


def _lib_digest():
    # the library is current when every object it was linked from is the one the sources now ask for
    return _digest(_link_cmd(), sources() + headers())


def _obj_stale(src):
    return not os.path.exists(_obj(src)) or _stamp_of(_obj(src)) != _obj_digest(src)


def needs_build():
    if not os.path.exists(OUT):
        return True
    stamp = _stamp_of(OUT)
    if stamp is None:
        # a library without a stamp (built by an older build.py, or copied to a box without csrc/build/): fall back to
        # modification times rather than demand a compiler where there may be none
        t = os.path.getmtime(OUT)
        return any(os.path.getmtime(p) > t for p in sources() + headers())
    return stamp != _lib_digest()


def _write_stamp(target, digest):
    tmp = target + ".stamp.tmp.%d" % os.getpid()
    with open(tmp, "w") as f:
        f.write(digest + "\n")
    os.replace(tmp, target + ".stamp")


def _audit(src, obj):
    from . import mfma_audit
    found = mfma_audit.audit_object(obj)
    if found:
        raise RuntimeError("%s: %d VALU write(s) of an MFMA's SrcC registers fewer than %d wait states behind it "
                           "(see image_matching_amd/mfma_audit.py):\n%s"
                           % (os.path.basename(src), len(found), mfma_audit.MIN_WAIT, mfma_audit.format_findings(found)))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)       # one builder at a time; the others find the work done
        if not force and not needs_build():
            return OUT
        todo = [s for s in sources() if force or _obj_stale(s)]

        def compile_one(src):
            obj = _obj(src)
            tmp = obj + ".tmp.%d" % os.getpid()
            cmd = _compile_cmd(src)
            cmd[-1] = tmp
            if verbose:
                print(" ".join(_compile_cmd(src)), flush=True)
            try:
                subprocess.check_call(cmd)
                if os.path.basename(src) in AUDITED:
                    _audit(src, tmp)
                os.replace(tmp, obj)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            _write_stamp(obj, _obj_digest(src))

        jobs = max(1, min(len(todo), int(os.environ.get("MAX_JOBS", "8")), os.cpu_count() or 1))
        if todo:
            with ThreadPoolExecutor(jobs) as ex:
                list(ex.map(compile_one, todo))
        cmd = _link_cmd()
        if verbose:
            print(" ".join(cmd), flush=True)
        tmp = OUT + ".tmp.%d" % os.getpid()
        cmd[cmd.index("-o") + 1] = tmp
        subprocess.check_call(cmd)
        os.replace(tmp, OUT)
        _write_stamp(OUT, _lib_digest())
    return OUT


if __name__ == "__main__":
    import sys
    build(force="--force" in sys.argv, verbose=True)
