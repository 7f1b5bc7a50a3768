"""Build libsrsran_phy_hip.so (gfx950 only) in-tree with hipcc.

    python -m srslte_amd.build            # build if sources are newer than the library
    python -m srslte_amd.build --force

The library is the product: hand-written HIP kernels + the C-ABI host layer (srslte_amd/csrc).
It is built IN-TREE (srslte_amd/lib/) so that it travels with the repository snapshot to the GPU box.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libsrsran_phy_hip.so")

SOURCES = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "tables", "*.h")) +
                 glob.glob(os.path.join(ROOT, "include", "srsran_amd", "*.h")))

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-result",
         "-Wno-unused-value", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


# per-source flags.  The FFT kernels are written with scalar f32 on purpose: on gfx950 v_pk_add/mul/fma_f32 have the same peak as v_fma_f32
# (64 flop per clock and SIMD) and cost more issue cycles, and the SLP vectoriser packs adjacent real / imaginary operations into them with a
# v_mov shuffle around every pack.  Measured with and without (one box): OFDM rx 415 -> 457 Gsamples/s (LTE), 397 -> 432 (NR); the register
# FFT of the PSS kernel 2.9 -> 2.8 ms before its other fixes.
_NO_PACK = ["-fno-slp-vectorize", "-mllvm", "-disable-vector-combine"]
EXTRA_FLAGS = {"pss_wave_kernels.hip": _NO_PACK, "ofdm_kernels.hip": _NO_PACK, "sync_kernels.hip": _NO_PACK}


# development only: extra compiler flags for timing experiments (e.g. -DLAT_EXP_NO_COLLECT), never set by the product build
_DEV_FLAGS = os.environ.get("SRSRAN_HIP_BUILD_FLAGS", "").split()


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS + [__file__])


def build(force=False, verbose=True, jobs=8):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        # an object newer than its source, every header and this file is kept (headers are shared: any header change rebuilds all)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(f) for f in [src, __file__] + HEADERS):
            continue
        cmd = [hipcc] + FLAGS + _DEV_FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if len(procs) >= jobs:
            _drain(procs)
    _drain(procs)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def _drain(procs):
    while procs:
        src, p = procs.pop(0)
        out, _ = p.communicate()
        if out.strip():
            print(out)
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s" % src)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
