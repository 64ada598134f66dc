"""Build libechopype_amd.so (HIP, gfx950 only) in-tree with hipcc.

    python echopype_amd/build.py [--force]      (run as a script: importing the package needs the
                                                 library this script produces)

The shared library is the product: there is no CPU fallback.  hipcc cross-compiles for gfx950
without a GPU, so this runs in the authoring container and the built .so travels with the tree.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIBDIR = os.path.join(HERE, "lib")
VARIANT = os.environ.get("EPA_VARIANT", "")  # tuning aid: separate objects + library per variant
LIBPATH = os.path.join(LIBDIR, f"libechopype_amd{('_' + VARIANT) if VARIANT else ''}.so")
OBJDIR = os.path.join(HERE, "csrc", "_obj" + (("_" + VARIANT) if VARIANT else ""))

SOURCES = ["runtime.hip", "power_coef.hip", "sv_power.hip", "block_reduce.hip", "fused_sv_mvbs.hip", "noise_apply.hip", "reduce_util.hip",
           "ek80_complex.hip", "ek80_fft.hip", "noise_masks.hip", "nasc.hip", "chain_fast.hip", "edge_exchange.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))
EXTRA = os.environ.get("EPA_EXTRA_FLAGS", "").split()
FLAGS = [*EXTRA, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical", f"-I{INCLUDE}", f"-I{CSRC}"]


def source_digest():
    """sha256 over every source the library is made of (csrc/*.hip, csrc/*.h, the public header) and the base compiler
    flags.  build_library() compiles it into the library (``epa_source_digest()``); ``_lib`` recomputes it from the
    files at import time and refuses a library built from other sources -- a stale shipped binary cannot pass."""
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, x) for x in HEADERS] + \
            [os.path.join(INCLUDE, "echopype_amd.h")]:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(f for f in FLAGS if not f.startswith("-I") and f not in EXTRA).encode())
    return h.hexdigest()[:32]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "echopype_amd.h"),
                                                       os.path.abspath(__file__)]
    jobs = []
    objs = []
    digest = source_digest()
    # the digest runtime.o was compiled with, written next to it when it is compiled (the directory is not tracked: a
    # checkout that changes a source cannot bring a matching stamp along and leave a stale runtime.o behind)
    stamp = os.path.join(OBJDIR, "runtime.o.digest")
    old = open(stamp).read().strip() if os.path.exists(stamp) else ""
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            raise FileNotFoundError(sp)
        op = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(op)
        # runtime.hip carries the digest of ALL sources: recompiled whenever any of them changed
        extra = [f'-DEPA_SOURCE_DIGEST="{digest}"'] if src == "runtime.hip" else []
        if force or _stale(op, [sp] + hdrs) or (extra and old != digest):
            jobs.append([hipcc, *FLAGS, *extra, "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if any("runtime.hip" in " ".join(j) for j in jobs):
        with open(stamp, "w") as f:
            f.write(digest + "\n")
    if jobs or force or _stale(LIBPATH, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIBPATH])
    return LIBPATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
