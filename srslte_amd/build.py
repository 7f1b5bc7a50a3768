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


# The measured-and-rejected kernel alternatives (turbo launch shapes "waves1" / "persistent", PSS correlation kernels "pair" / "block": DESIGN.md
# par. 3.2, 3.4) are NOT in the product library: the three kernel files that hold them are compiled a second time with SRSRAN_HIP_WITH_VARIANTS and
# linked with the product's other objects into tools/probe/lib/libsrsran_phy_hip_variants.so, which tests/test_gpu_variants.py and the variant
# measurements load through SRSRAN_HIP_LIB.
VARIANTS_LIB = os.path.join(ROOT, "tools", "probe", "lib", "libsrsran_phy_hip_variants.so")
VARIANT_SOURCES = ("turbo_kernels.hip", "pss_wave_kernels.hip", "sync_kernels.hip")


def _flags_stamp(objdir, flags):
    """objects are reused only when they were built by the same compiler with the same flags (a -D timing experiment must not leak into the product)"""
    import hashlib

    tag = hashlib.sha256(" ".join(flags).encode()).hexdigest()[:16]
    stamp = os.path.join(objdir, ".flags")
    same = os.path.exists(stamp) and open(stamp).read().strip() == tag
    if not same:
        for f in glob.glob(os.path.join(objdir, "*.o")):
            os.remove(f)
        with open(stamp, "w") as fh:
            fh.write(tag + "\n")
    return same


def build_variants(force=False, verbose=True, jobs=8):
    build(force=False, verbose=verbose, jobs=jobs)
    objs_product = [os.path.join(ROOT, "build", "obj", os.path.basename(s) + ".o") for s in SOURCES if os.path.basename(s) not in VARIANT_SOURCES]
    hipcc = os.environ.get("HIPCC", "hipcc")
    objdir = os.path.join(ROOT, "build", "obj_variants")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(VARIANTS_LIB), exist_ok=True)
    _flags_stamp(objdir, [hipcc] + FLAGS + ["-DSRSRAN_HIP_WITH_VARIANTS"])
    procs, objs = [], []
    for name in VARIANT_SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(objdir, name + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(f) for f in [src, __file__] + HEADERS):
            continue
        cmd = [hipcc] + FLAGS + ["-DSRSRAN_HIP_WITH_VARIANTS"] + EXTRA_FLAGS.get(name, []) + ["-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    _drain(procs)
    if force or not os.path.exists(VARIANTS_LIB) or any(os.path.getmtime(o) > os.path.getmtime(VARIANTS_LIB) for o in objs + objs_product):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", VARIANTS_LIB] + objs_product + objs + ["-ldl"])
    return VARIANTS_LIB


def build(force=False, verbose=True, jobs=8):
    hipcc = os.environ.get("HIPCC", "hipcc")
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    if not _flags_stamp(objdir, [hipcc] + FLAGS + _DEV_FLAGS + [k + "=" + " ".join(v) for k, v in sorted(EXTRA_FLAGS.items())]):
        force = True  # other flags than the objects in the tree were built with: everything is rebuilt
    if not force and not _stale():
        return LIB
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
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", LIB] + objs + ["-ldl"]
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
    # the variants library links the product's objects: once it exists it follows the product (a stale one lacks new entry points and the ctypes mirror
    # refuses to load it)
    if "--variants" in sys.argv or os.path.exists(VARIANTS_LIB):
        print(build_variants(force="--force" in sys.argv))
