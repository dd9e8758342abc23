"""Builds libannembed_hip.so (gfx950) in-tree with hipcc.  `python -m annembed_amd.build`

Incremental by CONTENT, not by mtime: `build_manifest.json` (next to the .so, git-ignored like it, travels to the GPU box with it) records
the sha256 of every translation unit together with the headers and flags it was compiled with.  A tree whose sources match the manifest and
whose .so exists compiles nothing -- also on a box that received the .so without the object files (`annembed_amd/build/` is in
.gpurunignore) or whose copy did not keep the mtimes.  `build()` prints a `build_mode:` line saying which of the two happened."""
import concurrent.futures as cf
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libannembed_hip.so")
MANIFEST = os.path.join(HERE, "build_manifest.json")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def _load_manifest():
    try:
        with open(MANIFEST) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def build(force=False, verbose=True):
    srcs = _sources()
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    headers.append(os.path.join(HERE, "..", "include", "annembed_hip.h"))
    common = hashlib.sha256((" ".join(FLAGS) + "|" + "|".join(_sha(h) for h in headers)).encode()).hexdigest()
    want = {os.path.basename(s): hashlib.sha256((common + _sha(s)).encode()).hexdigest() for s in srcs}
    have = _load_manifest().get("units", {})
    if not force and os.path.exists(OUT) and have == want:
        if verbose:
            print("build_mode: up to date (0 of %d units compiled; %s matches the sources by content)" % (len(srcs), os.path.basename(OUT)))
        return OUT
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or have.get(os.path.basename(s)) != want[os.path.basename(s)]:
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for s, r in ex.map(cc, jobs):
                if verbose and r.stderr.strip():
                    sys.stderr.write(r.stderr)
                if r.returncode != 0:
                    raise RuntimeError("hipcc failed on %s\n%s" % (s, r.stderr))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed\n" + r.stderr)
    with open(MANIFEST, "w") as f:
        json.dump({"units": want, "flags": FLAGS}, f, indent=0, sort_keys=True)
    if verbose:
        print("build_mode: compiled %d of %d units with %s and linked %s" % (len(jobs), len(srcs), os.path.basename(HIPCC), os.path.basename(OUT)))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
