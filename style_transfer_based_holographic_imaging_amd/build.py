"""In-tree build of libholo_hip.so: ``hipcc --offload-arch=gfx950 -shared`` over ``csrc/*.hip``.

No torch extension machinery: the boundary is a plain C-ABI (``include/holo_hip.h``) loaded with
ctypes, so the library builds without a GPU and travels to the GPU box inside the repo snapshot.

Staleness is decided by ``source_hash()`` — sources, headers and compiler flags — which is compiled into the
library (``holo_source_hash()``) and compared by ``_lib`` on every load; the ROCm version the object was compiled
with is stored beside it and makes ``needs_build()`` true when it differs, but is NOT part of the load-time check (a
library built on the build box must load on a GPU box with another ROCm patch level).
The compiler writes to a temporary file that is renamed over the library under a file lock, so
concurrent builders (pytest next to bench, xdist workers) neither clobber each other nor expose a
half-written object to ``dlopen``.
"""
from __future__ import annotations

import fcntl
import glob
import hashlib
import mmap
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libholo_hip.so")
LOCK = OUT + ".lock"
ARCH = "gfx950"
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "holo_hip.h")]


def _rocm_version():
    try:
        with open("/opt/rocm/.info/version") as f:
            return f.read().strip()
    except OSError:
        return "unknown"


def source_hash(extra_flags=()):
    """sha256 over the sources and headers and the compiler flags."""
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + list(extra_flags)).encode())
    return h.hexdigest()


def sources_present():
    """False for an install that ships the built library without csrc/ (nothing to hash, nothing to rebuild)."""
    return bool(sources()) and all(os.path.isfile(d) for d in _deps())


_TAG = b"@(#)holo_source_hash="
_ROCM_TAG = b"@(#)holo_rocm_version="


def _read_tag(path, tag):
    try:
        with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as m:
            i = m.find(tag)
            if i < 0:
                return None
            raw = m[i + len(tag): i + len(tag) + 64]
            return raw.split(b"\0")[0].decode("ascii", "replace")
    except (OSError, ValueError):
        return None


def built_rocm_version(path=OUT):
    return _read_tag(path, _ROCM_TAG)


def built_hash(path=OUT):
    """The hash compiled into a built library (what its holo_source_hash() returns), or None (missing / unreadable /
    older ABI).  Read from the file's bytes: loading it here would pin a stale image in this process (dlopen returns
    the already-loaded object for the same path after a rebuild)."""
    return _read_tag(path, _TAG)


def needs_build():
    """True when the library is missing or was built from other sources / flags, or by another ROCm release."""
    return built_hash() != source_hash() or built_rocm_version() != _rocm_version()


def _lock_path(out):
    """the lock file lives beside the library; a read-only package directory falls back to the temp dir"""
    path = LOCK if out == OUT else out + ".lock"
    if os.access(os.path.dirname(path) or ".", os.W_OK):
        return path
    import tempfile
    return os.path.join(tempfile.gettempdir(), "holo_hip_" + hashlib.sha256(os.path.abspath(out).encode()).hexdigest()[:16] + ".lock")


def build(force: bool = False, verbose: bool = True, extra_flags=(), out: str = OUT):
    """Build (if stale) and return the library path.  `extra_flags` / `out` make side builds
    (-DHOLO_DEBUG ablation knobs, -DHOLO_CONV_STAMPS=1) that never replace the shipped library."""
    out = os.path.abspath(out)                          # hipcc runs with csrc/ as its working directory
    want = source_hash(extra_flags)
    if not force and built_hash(out) == want and built_rocm_version(out) == _rocm_version():
        return out                                      # up to date: no lock file is touched (read-only installs)
    with open(_lock_path(out), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and built_hash(out) == want and built_rocm_version(out) == _rocm_version():     # re-checked under the lock: another process may just have built it
            return out
        tmp = f"{out}.tmp.{os.getpid()}"
        cmd = [_hipcc()] + FLAGS + list(extra_flags) + [f'-DHOLO_SOURCE_HASH="{want}"', f'-DHOLO_ROCM_VERSION="{_rocm_version()}"', "-o", tmp] + sources()
        if verbose:
            print("[holo build]", " ".join(cmd), flush=True)
        try:
            subprocess.run(cmd, check=True, cwd=CSRC)
            os.replace(tmp, out)
        finally:
            if os.path.exists(tmp):
                os.unlink(tmp)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
