"""shared by tests/test_gpu_ref_link.py (the reference's test programs on the product library) and tests/test_ref_link_oracle.py (the same
programs on the CPU oracle): data files in the formats those programs read, and the runner."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "ref_link", "_build")
GOLD = os.path.join(HERE, "golden")


def program(kind, name):
    """kind: "bin" (linked against libsrsran_phy_hip.so) or "bin_oracle" (linked against the oracle shim)"""
    p = os.path.join(BUILD, kind, name)
    if not os.path.exists(p):
        pytest.skip("tests/ref_link/_build/%s/%s not built (make -C tests/ref_link needs the reference tree: dev container)" % (kind, name))
    return p


def make_data_dir(d):
    """capture files + LDPC example files in the formats the reference programs read (from tests/golden, never from /root/reference)"""
    cap = np.load(os.path.join(GOLD, "sync_captures.npz"))
    cap["pbch_1_92M_x"].tofile(d / "signal.1.92M.dat")
    cap["amar_1_92M_sf0_x"].tofile(d / "signal.1.92M.amar.dat")
    cap["pcfich_10M_x"].tofile(d / "signal.10M.dat")
    z = np.load(os.path.join(GOLD, "ref_link_data.npz"))
    z["pmch_100prb_x"].tofile(d / "pmch_100prbs_MCS2_SR0.bin")
    for bg in (0, 1):
        with open(d / ("examplesBG%d.dat" % (bg + 1)), "w") as f:
            for b, Z in z["ldpc_sizes"]:
                if b != bg:
                    continue
                for part in ("msgs", "cwds"):
                    key = "bg%d_z%d_%s" % (bg, Z, part)
                    n = int(z[key + "_len"][0])
                    ones = np.unpackbits(z[key], axis=1)[:, :n]
                    fill = np.unpackbits(z[key + "_fill"], axis=1)[:, :n]
                    f.write("ls%d%s\n" % (Z, part))
                    for r in range(ones.shape[0]):
                        f.write("".join("-" if fill[r, i] else "01"[ones[r, i]] for i in range(n)) + "\n")
    return d


def run_program(kind, name, args, cwd, timeout=600):
    r = subprocess.run([program(kind, name)] + [str(a) for a in args], cwd=str(cwd), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, errors="replace", timeout=timeout)
    return r.returncode, r.stdout


def add_ctest_data(d):
    """the data files the reference's ctest lines name (tests/golden/ref_ctest_data.npz, written by tools/ref_ctest_manifest.py)"""
    z = np.load(os.path.join(GOLD, "ref_ctest_data.npz"))
    for i, name in enumerate(z["names"]):
        z["f%03d" % i].tofile(d / str(name))
    return d
