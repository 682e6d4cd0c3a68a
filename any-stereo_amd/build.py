"""Build libanystereo_hip.so (gfx950 only) in-tree with hipcc.

    python any-stereo_amd/build.py [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "anystereo", "lib")
LIB = os.path.join(LIBDIR, "libanystereo_hip.so")
HEADER = os.path.join(HERE, "..", "include", "anystereo_hip.h")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-pass-failed"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _newer(src_list, target) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def _source_hash() -> str:
    import importlib.util
    spec = importlib.util.spec_from_file_location("anystereo_srchash", os.path.join(HERE, "anystereo", "_srchash.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    # the hash of every native source, compiled into the library (as_source_hash): _lib.load() refuses a stale binary
    stamp_src = os.path.join(OBJ, "stamp.cpp")
    stamp = 'extern "C" const char* as_source_hash(void) { return "%s"; }\n' % _source_hash()
    if not os.path.exists(stamp_src) or open(stamp_src).read() != stamp:
        open(stamp_src, "w").write(stamp)
    deps = [os.path.join(CSRC, "common.h"), HEADER]
    hipcc = _hipcc()
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        if force or _newer([src] + deps, obj):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for (src, obj), r in ex.map(compile_one, jobs):
                if verbose and (r.stderr.strip() or r.returncode):
                    sys.stderr.write(r.stderr)
                if r.returncode != 0:
                    raise RuntimeError(f"hipcc failed on {src}")
                if verbose:
                    print(f"[build] compiled {os.path.basename(src)}")
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if force or jobs or _newer(objs + [stamp_src], LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [stamp_src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
