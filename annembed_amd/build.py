"""Builds libannembed_hip.so (gfx950) in-tree with hipcc.  `python -m annembed_amd.build`"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libannembed_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _needs(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = _sources()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "annembed_hip.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _needs(o, [s] + headers):
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
    if jobs or not os.path.exists(OUT):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed\n" + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
