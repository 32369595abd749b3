"""In-tree build of the HIP library (gfx950 only): hipcc -> safe-marl_amd/libflexenv_hip.so.
The .so is git-ignored but travels to the GPU box with the gpurun snapshot.

Staleness is decided by CONTENT, not mtime: the sha256 of every source / header / flag that goes into the library is
kept next to it (``libflexenv_hip.so.sha256``); ``build()`` recompiles whenever that digest differs from the sources in
the tree (a checkout, a copy to another box or a touched file cannot make a stale binary look fresh) and says which of
the two happened.  Translation units are compiled one by one into ``build/*.o`` (each keyed by its own digest) and
linked, so editing one kernel file recompiles one file."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SRC = [os.path.join(CSRC, f) for f in ("flexenv.hip", "actor.hip", "critic.hip", "rollout.hip", "wgrad.hip", "lnrelu.hip",
                                       "optim.hip", "tdloss.hip", "gru.hip")]
SRC = [f for f in SRC if os.path.exists(f)]
HEADERS = [os.path.join(CSRC, "flex_device.h"), os.path.join(CSRC, "flex_reduce.h"), os.path.join(CSRC, "flex_launch.h"),
           os.path.join(CSRC, "flex_td.h"), os.path.join(CSRC, "actor_r16.h"),
           os.path.join(ROOT, "include", "flexenv.h"), os.path.join(ROOT, "include", "flexnet.h")]
DEPS = SRC + HEADERS
OUT = os.path.join(HERE, "libflexenv_hip.so")
STAMP = OUT + ".sha256"
OBJ_DIR = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
FLAGS = CFLAGS + ["-shared"]          # kept for callers that build variants (tools/stamps.py)


def _sha(paths, extra=()):
    h = hashlib.sha256()
    for x in extra:
        h.update(str(x).encode())
        h.update(b"\0")
    for p in paths:
        h.update(os.path.basename(p).encode())
        h.update(b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def source_digest():
    """Digest of everything the library is made of: sources, headers and compiler flags."""
    return _sha(DEPS, extra=[a for a in CFLAGS if not a.startswith("-I")])


def built_digest():
    try:
        return open(STAMP).read().strip()
    except OSError:
        return None


def needs_build():
    return not os.path.exists(OUT) or built_digest() != source_digest()


def build(force=False, verbose=False):
    """Returns the library path.  ``build.last_action`` says what happened: "compiled" or "up to date"."""
    global last_action
    want = source_digest()
    if not force and os.path.exists(OUT) and built_digest() == want:
        last_action = "up to date"
        if verbose:
            print(f"[build] {os.path.basename(OUT)} is up to date (source digest {want[:16]})")
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs, jobs = [], []
    for src in SRC:
        key = _sha([src] + HEADERS, extra=[a for a in CFLAGS if not a.startswith("-I")])[:16]
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + "." + key + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj):
            cmd = [HIPCC] + CFLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((src, obj, subprocess.Popen(cmd)))
            if len(jobs) >= 4:                       # 8 CPUs / 64 GiB in the build container: a few hipcc at a time
                _wait(jobs)
    _wait(jobs)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(want + "\n")
    keep = set(objs)
    for f in os.listdir(OBJ_DIR):                    # objects of older source versions
        p = os.path.join(OBJ_DIR, f)
        if p.endswith(".o") and p not in keep:
            os.remove(p)
    last_action = "compiled"
    if verbose:
        print(f"[build] compiled {os.path.basename(OUT)} (source digest {want[:16]})")
    return OUT


def _wait(jobs):
    while jobs:
        src, obj, proc = jobs.pop(0)
        if proc.wait() != 0:
            if os.path.exists(obj):
                os.remove(obj)
            for _, o, p in jobs:
                p.wait()
            raise subprocess.CalledProcessError(proc.returncode, [HIPCC, src])


last_action = None

if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
