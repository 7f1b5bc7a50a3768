"""Lists the reference's own ctest lines for the PHY test programs that bind symbols of libsrsran_phy_hip.so (dev container only).

    python tools/ref_ctest_manifest.py        -> tests/ref_link/ctest_manifest.json
                                                 + tests/golden/ref_ctest_data.npz (the data files those lines name, as they are)

For every add_test / add_lte_test / add_nr_test / add_nbiot_test line with literal arguments in lib/src/phy/**/test/CMakeLists.txt the
manifest keeps: ctest name, program, argument list, the data files it names (by base name) and -- read from the linked program in
tests/ref_link/_build/bin_full with nm -- which product symbols it binds.  Lines whose program binds none are dropped (they would not
exercise the product), so are programs that need a radio.  The looped LDPC / PUSCH lines of the CMake files are represented by a
hand-picked literal subset at the end.  tests/test_gpu_ref_ctest.py runs every entry on the GPU box and expects exit code 0, which is
what ctest itself checks.
"""
import glob
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHY = "/root/reference/lib/src/phy"
BIN = os.path.join(ROOT, "tests", "ref_link", "_build", "bin_full")
LIB = os.path.join(ROOT, "srslte_amd", "lib", "libsrsran_phy_hip.so")
MAX_DATA = 320 * 1024  # bytes per data file kept as a fixture


def exported():
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip() and not ln.split()[-1].startswith("_")}


def bound(prog, exp):
    p = os.path.join(BIN, prog)
    if not os.path.exists(p):
        return None
    out = subprocess.check_output(["nm", "-D", "--undefined-only", p], text=True)
    return sorted({ln.split()[-1] for ln in out.splitlines() if ln.strip()} & exp)


def literal_lines():
    pat = re.compile(r"^\s*add_(?:lte_|nr_|nbiot_)?test\((.*)\)\s*(?:#.*)?$")
    for cm in sorted(glob.glob(PHY + "/*/test/CMakeLists.txt") + glob.glob(PHY + "/fec/*/test/CMakeLists.txt")):
        d = os.path.dirname(cm)
        for ln in open(cm):
            m = pat.match(ln)
            if not m:
                continue
            tok = m.group(1).split()
            if tok and tok[0] == "NAME":  # add_nr_test(NAME x COMMAND prog args...)
                if "COMMAND" not in tok:
                    continue
                name, tok = tok[1], tok[tok.index("COMMAND") + 1:]
                tok = [name] + tok
            if len(tok) < 2:
                continue
            name, prog, args = tok[0], tok[1], tok[2:]
            files = []
            ok = True
            for i, a in enumerate(args):
                mm = re.match(r"\$\{CMAKE_CURRENT_SOURCE_DIR\}/(.+)$", a) or re.match(r"\$\{CMAKE_HOME_DIRECTORY\}/lib/src/phy/[a-z_]+/test/(.+)$", a)
                if mm:
                    src = os.path.join(d, mm.group(1)) if "CURRENT" in a else os.path.join(PHY, a.split("lib/src/phy/")[1])
                    files.append((i, os.path.basename(src), src))
                elif "${" in a:
                    ok = False
            if ok:
                yield os.path.relpath(d, PHY), name, prog, args, files


EXTRA = [  # literal stand-ins for the looped lines (fec/ldpc/test/CMakeLists.txt:58-145,150-222; phch/test/CMakeLists.txt:470-545)
    ("fec/ldpc/test", "LDPC-DEC-BG1-LS%d", "ldpc_dec_test", "-b1 -l%d", (2, 36, 384), True),
    ("fec/ldpc/test", "LDPC-DEC-BG2-LS%d", "ldpc_dec_test", "-b2 -l%d", (9, 208), True),
    ("fec/ldpc/test", "LDPC-DEC-S-BG1-LS%d", "ldpc_dec_s_test", "-b1 -l%d", (2, 36, 384), True),
    ("fec/ldpc/test", "LDPC-DEC-S-BG2-LS%d", "ldpc_dec_s_test", "-b2 -l%d", (9, 208), True),
    ("fec/ldpc/test", "LDPC-DEC-AVX2-BG1-LS%d", "ldpc_dec_avx2_test", "-b1 -l%d", (2, 36, 384), True),
    ("fec/ldpc/test", "LDPC-DEC-AVX2-BG2-LS%d", "ldpc_dec_avx2_test", "-b2 -l%d", (15, 384), True),
    ("fec/ldpc/test", "LDPC-DEC-AVX2-FLOOD-BG1-LS%d", "ldpc_dec_avx2_test", "-b1 -l%d -x1", (384,), True),
    ("fec/ldpc/test", "LDPC-ENC-BG1-LS%d", "ldpc_enc_test", "-b1 -l%d", (2, 36, 384), True),
    ("fec/ldpc/test", "LDPC-ENC-BG2-LS%d", "ldpc_enc_test", "-b2 -l%d", (9, 208, 384), True),
    ("fec/ldpc/test", "LDPC-ENC-AVX2-BG1-LS%d", "ldpc_enc_avx2_test", "-b1 -l%d", (36, 384), True),
    ("fec/ldpc/test", "LDPC-ENC-AVX2-BG2-LS%d", "ldpc_enc_avx2_test", "-b2 -l%d", (15, 208), True),
    ("fec/ldpc/test", "LDPC-RM-b1-l%d-r0", "ldpc_rm_test", "-b1 -l%d -e%d -f10 -m2 -r0 -M%d", ((8, 528, 528), (256, 16896, 16896), (256, 8448, 8448)), False),
    ("fec/ldpc/test", "LDPC-RM-b2-l%d-r2", "ldpc_rm_test", "-b2 -l%d -e%d -f10 -m3 -r2 -M%d", ((16, 798, 800), (128, 12798, 3200), (64, 1596, 3200)), False),
    ("fec/ldpc/test", "LDPC-RM-b1-l%d-r3", "ldpc_rm_test", "-b1 -l%d -e%d -f10 -m4 -r3 -M%d", ((32, 2112, 2112), (128, 16896, 4224)), False),
    ("phch/test", "pusch_test-n%d-L%d-m%d", "pusch_test", "-n %d -L %d -m %d", ((6, 6, 0), (25, 25, 14), (50, 50, 21), (100, 50, 7)), False),
    ("phch/test", "pusch_test-n%d-L%d-m%d-ack-cqi", "pusch_test", "-n %d -L %d -m %d -p uci_ack 2 -p cqi wideband", ((50, 50, 14), (100, 50, 21)), False),
    ("phch/test", "pusch_test-n%d-L%d-m%d-ack-cqi-64qam", "pusch_test", "-n %d -L %d -m %d -p uci_ack 2 -p cqi wideband -p enable_64qam", ((100, 50, 27),), False),
]


def phy_level_lines():
    """lib/test/phy/CMakeLists.txt: the eNB-DL -> UE-DL loop-back program of BASELINE configs[0] ("CPU reference via lib/test/phy").  Its 240
    phy_dl_test lines come from nested foreach loops (:33-59), restated here; pucch_ca_test (:63) and the five phy_dl_nr_test lines (:67-74) are literal."""
    for prb in (6, 15, 25, 50, 75, 100):
        for q in (0, 1):
            for tm in (1, 2, 3, 4):
                for mcs in range(0, 29, 7):
                    if q and mcs == 28:  # :42-48: with 256-QAM tables the top index is 27 (26 at 15 PRB)
                        mcs = 26 if prb == 15 else 27
                    args = ["-p", str(prb), "-t", str(tm)] + (["-q"] if q else []) + ["-m", str(mcs)]
                    yield "../../test/phy", "phy_dl_test" + "".join(args), "phy_dl_test", args
    yield "../../test/phy", "pucch_ca_test", "pucch_ca_test", []
    yield "../../test/phy", "phy_dl_nr_test", "phy_dl_nr_test", "-p 100 -m 28".split()
    yield ("../../test/phy", "phy_dl_nr_test_rvd", "phy_dl_nr_test",
           "-P 52 -p 52 -m 0 -R 0 52 1 010010010010 00000000010000 -R 0 52 1 100100100100 00000010000000".split())
    yield "../../test/phy", "phy_dl_nr_test_cfo_delay", "phy_dl_nr_test", "-P 52 -p 52 -m 27 -C 100.0 -D 4 -n 10".split()
    yield "../../test/phy", "phy_dl_nr_test_52prb", "phy_dl_nr_test", "-P 52 -p 52 -m 27 -T 256qam -v -d 1 1 -n 10".split()
    yield "../../test/phy", "phy_dl_nr_test_270prb", "phy_dl_nr_test", "-P 270 -p 270 -m 27 -T 256qam -v -d 1 1 -n 10".split()


def main():
    exp = exported()
    entries, data = [], {}
    seen = set()
    for d, name, prog, args, files in literal_lines():
        if prog.endswith("_usrp") or (d, name, tuple(args)) in seen:
            continue
        seen.add((d, name, tuple(args)))
        b = bound(prog, exp)
        if not b:
            continue
        big = [f for f in files if os.path.getsize(f[2]) > MAX_DATA]
        if big:
            continue
        for _, base, src in files:
            data[base] = src
        args = [("@" + files[[f[0] for f in files].index(i)][1]) if i in [f[0] for f in files] else a for i, a in enumerate(args)]
        entries.append({"dir": d, "name": name, "program": prog, "args": args, "binds": len(b), "binds_sample": b[:6]})
    for d, name, prog, fmt, vals, ldpc_file in EXTRA:
        b = bound(prog, exp)
        if not b:
            continue
        for v in vals:
            v = v if isinstance(v, tuple) else (v,)
            entries.append({"dir": d, "name": name % v[:name.count("%d")], "program": prog, "args": (fmt % v).split(), "binds": len(b),
                            "binds_sample": b[:6], "needs_ldpc_examples": bool(ldpc_file)})
    for d, name, prog, args in phy_level_lines():
        b = bound(prog, exp)
        if b:
            entries.append({"dir": d, "name": name, "program": prog, "args": args, "binds": len(b), "binds_sample": b[:6]})
    out = os.path.join(ROOT, "tests", "ref_link", "ctest_manifest.json")
    json.dump({"entries": entries, "data_files": sorted(data)}, open(out, "w"), indent=1)
    import numpy as np

    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_ctest_data.npz"),
                        **{"f%03d" % i: np.fromfile(data[k], dtype=np.uint8) for i, k in enumerate(sorted(data))}, names=np.array(sorted(data)))
    progs = sorted({e["program"] for e in entries})
    print("%d test lines, %d programs, %d data files (%d KB)" % (len(entries), len(progs), len(data), sum(os.path.getsize(s) for s in data.values()) // 1024))
    print(" ".join(progs))


if __name__ == "__main__":
    main()
