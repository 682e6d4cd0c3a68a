"""Content hash of the native sources (csrc/*.hip, csrc/*.h, include/anystereo_hip.h).  build.py compiles it into the library
(`as_source_hash()`), `_lib.load()` recomputes it from the tree and refuses a library built from other sources: the built `.so`
is git-ignored but travels with the repo snapshot, so a stale binary would otherwise run silently on the GPU box."""
import hashlib
import os

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "..", "csrc")
HEADER = os.path.join(PKG, "..", "..", "include", "anystereo_hip.h")


def source_hash() -> str | None:
    """None when the sources are not there (an installed library without its tree)."""
    if not (os.path.isdir(CSRC) and os.path.exists(HEADER)):
        return None
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(HEADER, "rb").read())
    return h.hexdigest()[:16]
