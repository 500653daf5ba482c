"""Builds sparse2dense_amd/lib/libs2d_hip.so (hipcc, gfx950 only) from sparse2dense_amd/csrc/*.hip.

    python -m sparse2dense_amd.build [--force]

The .so is built IN-TREE so that it travels with the repo snapshot to the GPU box."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libs2d_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-fast-math", "-ffp-contract=off"]


# per-file extra flags; the flags an object was built with are kept beside it (<name>.flags, written only after hipcc succeeded) and are
# part of the staleness test.
# losses.hip is built WITHOUT the SLP vectoriser, i.e. without packed-FP32 code: `pcr_level_bwd_dense_kernel`'s compiler-generated
# `v_pk_fma_f32` returned different high-lane sums while another queue's MFMA kernel shared the SIMDs (DESIGN rule 36), and the default
# graphed mode does run weight-gradient MFMA kernels on a second queue.  Costs 0.1 ms per step.  S2D_BUILD_LOSSES_SLP=1 restores the
# vectorised build for A/B runs.
EXTRA = {}
if os.environ.get("S2D_BUILD_LOSSES_SLP") != "1":
    EXTRA["losses.hip"] = ["-fno-slp-vectorize"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(os.path.dirname(HERE), "include", "s2d.h")]
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        flags = FLAGS + EXTRA.get(os.path.basename(s), [])
        stamp = o[:-2] + ".flags"
        built_with = open(stamp).read() if os.path.exists(stamp) else " ".join(FLAGS)   # (objects from before the stamps: the default flags)
        if force or _stale(o, [s] + hdrs) or built_with != " ".join(flags):
            for stale in (o, stamp):   # a failed or interrupted compile must not leave an object that looks current
                if os.path.exists(stale):
                    os.remove(stale)
            cmd = [HIPCC] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr, flush=True)
            procs.append((s, subprocess.Popen(cmd), stamp, " ".join(flags)))
    failed = []
    for s, p, stamp, flag_line in procs:
        if p.wait() != 0:
            failed.append(s)
            continue
        with open(stamp, "w") as f:
            f.write(flag_line)
    if failed:
        raise RuntimeError(f"hipcc failed on {failed}")
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr, flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
