"""Builds libsgo_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

One object per translation unit, compiled in parallel, then one link.  Freshness is decided by CONTENT, not by
mtime: every object carries the SHA-256 of its source, of every header under csrc/ and include/, and of the flags;
the library carries the hashes of its objects (libsgo_hip.so.manifest.json).  `build_lib()` returns a report saying
what was compiled, so a caller (and __graft_entry__.build()) can state whether the shipped binary matches the
sources or was rebuilt."""
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(HERE, "..", "include")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libsgo_hip.so")
MANIFEST = LIB + ".manifest.json"
SOURCES = ["sgo_rules.hip", "sgo_engine.hip", "sgo_conv.hip"]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
          "-Wno-unused-function"] + os.environ.get("SGO_EXTRA_CFLAGS", "").split()
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _headers():
    hs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hpp", ".h"))]
    hs += [os.path.join(INCLUDE, f) for f in sorted(os.listdir(INCLUDE)) if f.endswith(".h")]
    return hs


def source_hashes():
    """{source file: content hash of the source, all headers and the compile flags}."""
    hs = _headers()
    return {s: _sha([os.path.join(CSRC, s)] + hs, " ".join(CFLAGS)) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))}


def _read_manifest():
    try:
        with open(MANIFEST) as f:
            return json.load(f)
    except Exception:
        return {}


def is_fresh():
    """True when libsgo_hip.so exists and its manifest matches the current sources."""
    m = _read_manifest()
    return os.path.exists(LIB) and m.get("sources") == source_hashes() and m.get("lib_sha256") == _sha([LIB])


def build_lib(force=False, verbose=False):
    """Compiles what is stale and links.  Returns {"lib": path, "compiled": [sources], "linked": bool}."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    want = source_hashes()
    report = {"lib": LIB, "compiled": [], "linked": False}
    if not force and is_fresh():
        return report
    os.makedirs(OBJDIR, exist_ok=True)
    jobs = []
    for src, digest in want.items():
        obj = os.path.join(OBJDIR, src + ".o")
        tag = obj + ".sha256"
        have = open(tag).read().strip() if os.path.exists(tag) and os.path.exists(obj) else None
        if force or have != digest:
            jobs.append((src, obj, tag, digest))

    def compile_one(job):
        src, obj, tag, digest = job
        cmd = [hipcc] + CFLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        with open(tag, "w") as f:
            f.write(digest)
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 4)) as ex:
            report["compiled"] = list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJDIR, s + ".o") for s in want]
    cmd = [hipcc] + LDFLAGS + ["-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    report["linked"] = True
    with open(MANIFEST, "w") as f:
        json.dump({"sources": want, "lib_sha256": _sha([LIB]), "flags": CFLAGS}, f, indent=1, sort_keys=True)
    return report


if __name__ == "__main__":
    r = build_lib(force="--force" in sys.argv, verbose=True)
    print(json.dumps(r))
