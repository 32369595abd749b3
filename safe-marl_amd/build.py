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
                                       "optim.hip", "tdloss.hip", "gru.hip", "linear.hip", "opf.hip")]
SRC = [f for f in SRC if os.path.exists(f)]
# every header under csrc/ and include/ (a list by name missed critic_finish.h and window_refresh.h when they were added:
# an edit to either neither rebuilt the library nor changed the digest the profiles are stamped with)
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
          sorted(os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h"))
DEPS = SRC + HEADERS
OUT = os.path.join(HERE, "libflexenv_hip.so")
STAMP = OUT + ".sha256"
OBJ_DIR = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
# the compiler's own per-kernel resource report (registers, scratch, LDS, occupancy): parsed into kernel_resources.json next
# to the library, so that bench.py quotes the registers / waves per SIMD of the build it is timing
REMARKS = ["-Rpass-analysis=kernel-resource-usage"]
RESOURCES = os.path.join(HERE, "kernel_resources.json")
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
    if not force and os.path.exists(OUT) and built_digest() == want and os.path.exists(RESOURCES):
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
        if force or not os.path.exists(obj) or not os.path.exists(obj + ".log"):
            cmd = [HIPCC] + CFLAGS + REMARKS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((src, obj, subprocess.Popen(cmd, stderr=open(obj + ".log", "w"))))
            if len(jobs) >= 4:                       # 8 CPUs / 64 GiB in the build container: a few hipcc at a time
                _wait(jobs)
    _wait(jobs)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(want + "\n")
    _write_resources(objs, want)
    keep = set(objs) | {o + ".log" for o in objs}
    for f in os.listdir(OBJ_DIR):                    # objects of older source versions
        p = os.path.join(OBJ_DIR, f)
        if (p.endswith(".o") or p.endswith(".o.log")) and p not in keep:
            os.remove(p)
    last_action = "compiled"
    if verbose:
        print(f"[build] compiled {os.path.basename(OUT)} (source digest {want[:16]})")
    return OUT


_FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
           "Occupancy [waves/SIMD]": "waves_per_simd", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills",
           "LDS Size [bytes/block]": "lds_bytes_per_block"}


def _parse_remarks(text):
    """{mangled kernel name: {vgprs, agprs, ...}} from -Rpass-analysis=kernel-resource-usage output."""
    out, cur = {}, None
    for line in text.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].split("[-Rpass-analysis", 1)[0].strip()
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            if k.strip() in _FIELDS:
                try:
                    cur[_FIELDS[k.strip()]] = int(v)
                except ValueError:
                    pass
    return out


def _write_resources(objs, digest):
    import json
    kernels = {}
    for o in objs:
        try:
            kernels.update(_parse_remarks(open(o + ".log").read()))
        except OSError:
            pass
    names = list(kernels)
    try:                                             # readable names next to the mangled ones
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    except Exception:
        dem = names
    for n, d in zip(names, dem):
        kernels[n]["name"] = d.split("(")[0].replace("void ", "", 1) if d != n else n
    with open(RESOURCES, "w") as f:
        json.dump({"source_digest": digest, "kernels": kernels}, f, indent=1, sort_keys=True)


def kernel_resources(name_prefix=None):
    """The compiler's resource report of the library as built: {mangled: {name, vgprs, agprs, sgprs, waves_per_simd,
    scratch_bytes_per_lane, lds_bytes_per_block, ...}}; ``name_prefix`` filters on the demangled name.  {} when the report
    is missing or belongs to another build."""
    import json
    try:
        r = json.load(open(RESOURCES))
    except (OSError, ValueError):
        return {}
    if r.get("source_digest") != built_digest():
        return {}
    k = r.get("kernels", {})
    if name_prefix is not None:
        k = {m: v for m, v in k.items() if v.get("name", "").startswith(name_prefix)}
    return k


def _wait(jobs):
    while jobs:
        src, obj, proc = jobs.pop(0)
        if proc.wait() != 0:
            try:
                sys.stderr.write(open(obj + ".log").read()[-4000:])
            except OSError:
                pass
            if os.path.exists(obj):
                os.remove(obj)
            for _, o, p in jobs:
                p.wait()
            raise subprocess.CalledProcessError(proc.returncode, [HIPCC, src])


last_action = None

if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
