"""In-tree build of libholo_hip.so: ``hipcc --offload-arch=gfx950 -shared`` over ``csrc/*.hip``.

No torch extension machinery: the boundary is a plain C-ABI (``include/holo_hip.h``) loaded with
ctypes, so the library builds without a GPU and travels to the GPU box inside the repo snapshot.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libholo_hip.so")
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "holo_hip.h")]


def _source_hash():
    import hashlib
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


HASH_FILE = OUT + ".srchash"


def needs_build():
    """True when the library is missing or was built from other sources.  Decided by a content hash written next to the
    library (file times do not survive the copy to the GPU box); without the hash file, by modification times."""
    if not os.path.exists(OUT):
        return True
    if os.path.exists(HASH_FILE):
        with open(HASH_FILE) as f:
            return f.read().strip() != _source_hash()
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False, verbose: bool = True):
    if not force and not needs_build():
        return OUT
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", "-o", OUT] + sources()
    if verbose:
        print("[holo build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    with open(HASH_FILE, "w") as f:
        f.write(_source_hash() + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
