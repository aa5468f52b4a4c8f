"""Builds libmogan_hip.so (gfx950) from csrc/*.hip with hipcc, in-tree.  hipcc cross-compiles
without a GPU; the .so is git-ignored but travels to the GPU box with the snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmogan_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value"]


EXTRA = os.environ.get("MOGAN_CFLAGS", "").split()       # e.g. -DMOGAN_X6=0: the native fp32-MFMA form of every kernel


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdr = os.path.join(os.path.dirname(HERE), "include", "mogan_hip.h")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    # the flag set is part of the build's identity (MOGAN_CFLAGS=-DMOGAN_X6=0 switches the arithmetic form of every
    # kernel): objects compiled with other flags are stale whatever their mtimes say
    stamp = os.path.join(objdir, "flags.stamp")
    flagline = " ".join(FLAGS + EXTRA)
    if not os.path.exists(stamp) or open(stamp).read() != flagline:
        force = True
        with open(stamp, "w") as f:
            f.write(flagline)
    srcs = sources()
    hdrs = [hdr] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]

    def cc(pair):
        src, obj = pair
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + EXTRA + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, zip(srcs, objs)))
    if force or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
