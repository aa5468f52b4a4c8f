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


OUT_F32 = os.path.join(HERE, "libmogan_hip_f32.so")


def _build_variant(out, objdir, extra, force=False, verbose=True):
    hdr = os.path.join(os.path.dirname(HERE), "include", "mogan_hip.h")
    os.makedirs(objdir, exist_ok=True)
    # the flag set is part of the build's identity (-DMOGAN_X6=0 switches the arithmetic form of every kernel): objects
    # compiled with other flags are stale whatever their mtimes say
    stamp = os.path.join(objdir, "flags.stamp")
    flagline = " ".join(FLAGS + extra)
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
            cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=int(os.environ.get("MOGAN_BUILD_JOBS", "6"))) as ex:
        list(ex.map(cc, zip(srcs, objs)))
    if force or _stale(out, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


def build(force=False, verbose=True):
    """libmogan_hip.so: the product library (split-bf16 form of the MFMA kernels unless MOGAN_CFLAGS says otherwise)."""
    return _build_variant(OUT, os.path.join(HERE, "build"), EXTRA, force, verbose)


def build_f32(force=False, verbose=True):
    """libmogan_hip_f32.so: the same sources with -DMOGAN_X6=0, i.e. every MFMA kernel on the native v_mfma_f32_32x32x2_f32.
    Not loaded by the product; it is the reference point of the precision claims (tools/diag_x6_precision.py) and is kept
    under test by tests/test_kernels_gpu.py::test_native_fp32_mfma_build (MOGAN_LIB selects it)."""
    return _build_variant(OUT_F32, os.path.join(HERE, "build_f32"), ["-DMOGAN_X6=0"], force, verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
    if "--f32" in sys.argv:
        build_f32(force="--force" in sys.argv)
        print(OUT_F32)
