"""Builds gst_tacotron_amd/lib/libgsttaco.so (C-ABI + gfx950 kernels) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU; the built .so travels with the tree.
``python -m gst_tacotron_amd.build [--force]``
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgsttaco.so")
SOURCES = ["gsttaco.cpp", "skinny_gemm.hip", "gemm_conv.hip", "conv_wino_split.hip", "attention.hip", "dec_front.hip", "dec_front_lsa.hip", "persist_decode.hip", "gst.hip", "audio.hip"]
FLAGS_STAMP = os.path.join(LIBDIR, ".flags")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _headers():
    """Every header a source may include: all of csrc/*.h plus the public ABI header.  Globbed, so a new header cannot be
    forgotten (an edit to the hottest header once rebuilt nothing: a stale .so then ships to the GPU box)."""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.normpath(os.path.join(HERE, "..", "include", "gsttaco.h"))]


def _flags_changed():
    """True when the compile flags / compiler differ from the ones the objects in lib/ were built with."""
    want = " ".join([_hipcc()] + FLAGS)
    try:
        return open(FLAGS_STAMP).read() != want
    except OSError:
        return True


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


LAST_BUILD = {"compiled": [], "linked": False, "mode": "not run"}     # what the last build() call actually did (reported by __graft_entry__.build)


def build(force=False, verbose=False, debug=False):
    """debug=True adds -DGSTTACO_DEBUG: the experiment knobs are compiled in.  Never the default."""
    global FLAGS
    if debug and "-DGSTTACO_DEBUG" not in FLAGS:
        FLAGS = FLAGS + ["-DGSTTACO_DEBUG"]
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = _headers()
    if _flags_changed():
        force = True
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs):
            jobs.append([_hipcc()] + FLAGS + ["-x", "hip", "-c", sp, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn)
    linked = bool(force or jobs or _stale(LIB, objs))
    if linked:
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    LAST_BUILD["compiled"] = [os.path.basename(j[-3]) for j in jobs]
    LAST_BUILD["linked"] = linked
    LAST_BUILD["mode"] = ("compiled all sources" if len(jobs) == len(SOURCES) else "compiled %d of %d sources" % (len(jobs), len(SOURCES)) if jobs
                          else "relinked" if linked else "reused the up-to-date in-tree library")
    with open(FLAGS_STAMP, "w") as f:
        f.write(" ".join([_hipcc()] + FLAGS))
    return LIB


def describe():
    """One line saying which binary a process is about to load, WITHOUT compiling anything: whether the in-tree library is up to date
    against the sources lying beside it, its hash and size.  ``__graft_entry__.smoke()`` prints it, so the record of a GPU run says
    which build it tested (the GPU box runs the library cross-compiled in the build container)."""
    import hashlib
    if not os.path.exists(LIB):
        return "build_mode: NO LIBRARY at {} (python -m gst_tacotron_amd.build)".format(LIB)
    hdrs = _headers()
    stale = [src for src in SOURCES if _stale(os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o"), [os.path.join(CSRC, src)] + hdrs)]
    objs = [os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o") for src in SOURCES]
    state = ("prebuilt in-tree library, up to date against csrc/ (nothing compiled in this process)" if not stale and not _stale(LIB, [o for o in objs if os.path.exists(o)])
             else "prebuilt in-tree library, OLDER than " + ", ".join(stale or ["its objects"]))
    data = open(LIB, "rb").read()
    return "build_mode: {}; libgsttaco.so sha256[:16] {} ({} bytes){}".format(
        state, hashlib.sha256(data).hexdigest()[:16], len(data), "; flags changed since the build" if _flags_changed() else "")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, debug="--debug" in sys.argv))
