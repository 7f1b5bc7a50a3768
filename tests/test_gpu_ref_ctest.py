"""The reference's ctest lines for its PHY test programs, run on top of the product library.

tests/ref_link/Makefile target `ext` links EVERY test program of lib/src/phy/**/test that builds against libsrsran_phy_hip.so with the WHOLE
table of INTEGRATION.md section 1 removed from the reference side (OFDM, DFT, transform precoding, turbo decoder + encoder, QPP, code-block
segmentation, LDPC decoder + encoder + rate matcher, base graphs, PSS, SSS, sync, CFO, CP, soft demodulator).  tests/ref_link/ctest_manifest.json
(tools/ref_ctest_manifest.py) lists the ctest lines of the reference's CMake files for the programs that bind product symbols; ctest's
criterion is the exit code, and so is this test's.  Programs cover the LTE / NB-IoT / sidelink / NR channels whose objects call the
replaced entry points (PDSCH / PUSCH / PMCH through sch.c and the turbo coder pair, PRACH and channel estimation through srsran_dft_*,
NR PDSCH / PUSCH through sch_nr.c and the LDPC objects, NPSS / NSSS / PSSS through the DFT plans, ...).

By default one line per program plus every line of the programs on the hot path proper runs (a few minutes of process start-ups);
REF_CTEST_FULL=1 runs all of them (profiles/r03_ref_ctest_full.log is such a run).
"""
import json
import os

import pytest
from ref_link_common import HERE, add_ctest_data, make_data_dir, run_program

pytestmark = pytest.mark.gpu

MANIFEST = json.load(open(os.path.join(HERE, "ref_link", "ctest_manifest.json")))["entries"]
ALL_LINES_OF = {"ofdm_test", "dft_test", "turbodecoder_test", "turbocoder_test", "rm_turbo_test", "sync_test", "cfo_test", "ldpc_chain_test",
                "ldpc_rm_chain_test", "ldpc_dec_test", "ldpc_dec_s_test", "ldpc_dec_avx2_test", "ldpc_enc_test", "ldpc_enc_avx2_test",
                "ldpc_rm_test", "pusch_test", "sch_nr_test", "pusch_nr_test", "pdsch_nr_test", "pmch_test", "pbch_file_test", "pcfich_file_test",
                "phich_file_test", "pdcch_file_test", "pdsch_pdcch_file_test", "pmch_file_test", "modem_test", "prach_test", "phy_dl_nr_test", "pucch_ca_test"}


def _phy_dl_default(e):
    """lib/test/phy/phy_dl_test (eNB-DL -> UE-DL loop-back, BASELINE configs[0]'s harness): every 6-PRB line (N=128, the configs[0] cell) and the
    top-MCS lines of the 100-PRB cell (configs[1]'s cell: 13 code blocks of 6144 per transport block) by default, all 240 with REF_CTEST_FULL=1"""
    a = e["args"]
    prb, mcs = int(a[a.index("-p") + 1]), int(a[a.index("-m") + 1])
    return prb == 6 or (prb == 100 and mcs >= 27)


def _selected():
    full = os.environ.get("REF_CTEST_FULL", "0") == "1"
    seen = {}
    for e in MANIFEST:
        n = seen.get(e["program"], 0)
        seen[e["program"]] = n + 1
        if e["program"] == "phy_dl_test":
            if full or _phy_dl_default(e):
                yield e
        elif full or e["program"] in ALL_LINES_OF or n % 8 == 0:
            yield e


@pytest.fixture(scope="module")
def data_dir(tmp_path_factory, hiplib):
    return add_ctest_data(make_data_dir(tmp_path_factory.mktemp("ref_ctest_data")))


@pytest.mark.parametrize("entry", list(_selected()), ids=lambda e: "%s:%s" % (e["program"], e["name"]))
def test_ctest_line(entry, data_dir):
    args = [str(data_dir / a[1:]) if a.startswith("@") else a for a in entry["args"]]
    rc, out = run_program("bin_full", entry["program"], args, data_dir, timeout=900)
    assert rc == 0, "%s %s -> %d\n%s" % (entry["program"], " ".join(args), rc, out[-3000:])


# ---- the transport-block seam (tests/ref_link/Makefile target `tb`, tests/ref_link/tb_bind.c): the programs whose link contains the reference's
# sch.c, built a second time with ITS decode_tb_cb (sch.c:370) weakened in the object file and the library's in its place -- every transport
# block that srsran_pdsch_decode / srsran_pusch_decode / srsran_pmch_decode hand to srsran_dlsch_decode2 / srsran_ulsch_decode goes to the
# device as one call (rate de-matching, turbo early stop, CRCs), on the reference's own soft-buffer structs.
# NR (tests/ref_link/nr_bind.c): srsran_dlsch_nr_decode / srsran_ulsch_nr_decode (sch_nr.c:724-749) of the unmodified sch_nr.o weakened, the
# binding's in their place -- sch_nr_test drives them directly, pdsch_nr_test / pusch_nr_test through srsran_pdsch_nr_decode / srsran_pusch_nr_decode
TB_PROGRAMS = {"pdsch_test", "pusch_test", "pmch_test", "pdsch_pdcch_file_test", "pmch_file_test", "sch_nr_test", "pdsch_nr_test", "pusch_nr_test",
               "phy_dl_test", "phy_dl_nr_test"}


def _selected_tb():
    full = os.environ.get("REF_CTEST_FULL", "0") == "1"
    seen = {}
    for e in MANIFEST:
        if e["program"] not in TB_PROGRAMS:
            continue
        n = seen.get(e["program"], 0)
        seen[e["program"]] = n + 1
        if e["program"] == "phy_dl_test":
            if full or _phy_dl_default(e):
                yield e
        elif full or e["program"] != "pdsch_test" or n % 4 == 0:
            yield e


@pytest.mark.parametrize("entry", list(_selected_tb()), ids=lambda e: "tb:%s:%s" % (e["program"], e["name"]))
def test_ctest_line_through_the_transport_block_seam(entry, data_dir):
    args = [str(data_dir / a[1:]) if a.startswith("@") else a for a in entry["args"]]
    rc, out = run_program("bin_tb", entry["program"], args, data_dir, timeout=900)
    assert rc == 0, "%s %s -> %d\n%s" % (entry["program"], " ".join(args), rc, out[-3000:])


# ---- the grant-level seam (tests/ref_link/Makefile target `chan`, tests/ref_link/chan_bind.c): on top of the transport-block seam, srsran_pusch_decode,
# srsran_pdsch_decode, srsran_pdsch_encode and srsran_ulsch_encode of the unmodified pusch.o / pdsch.o / sch.o renamed to <name>_ref and the binding in
# their place: a grant the device path takes is ONE device call (include/srsran_amd/phy_chan_abi.h); UCI on PUSCH, several ports / codewords on the PDSCH
# receive side fall through to the renamed originals.  CHAN_BIND_REPORT=1 makes the binding print how many grants went which way.
CHAN_PROGRAMS = {"pusch_test", "pdsch_test", "phy_dl_test", "pdsch_pdcch_file_test"}  # (pmch_test reaches none of the four functions)


def _selected_chan():
    full = os.environ.get("REF_CTEST_FULL", "0") == "1"
    seen = {}
    for e in MANIFEST:
        if e["program"] not in CHAN_PROGRAMS:
            continue
        n = seen.get(e["program"], 0)
        seen[e["program"]] = n + 1
        if e["program"] == "phy_dl_test":
            if full or _phy_dl_default(e):
                yield e
        elif full or e["program"] != "pdsch_test" or n % 4 == 0:
            yield e


def _device_share(out):
    import re

    m = re.search(r"\[chan_bind\] pusch_decode dev (\d+) ref (\d+) \| pdsch_decode dev (\d+) ref (\d+) \| pdsch_encode dev (\d+) ref (\d+) \| ulsch_encode dev (\d+) ref (\d+)", out)
    return [int(v) for v in m.groups()] if m else None


@pytest.mark.parametrize("entry", list(_selected_chan()), ids=lambda e: "chan:%s:%s" % (e["program"], e["name"]))
def test_ctest_line_through_the_grant_seam(entry, data_dir):
    args = [str(data_dir / a[1:]) if a.startswith("@") else a for a in entry["args"]]
    os.environ["CHAN_BIND_REPORT"] = "1"
    try:
        rc, out = run_program("bin_chan", entry["program"], args, data_dir, timeout=900)
    finally:
        del os.environ["CHAN_BIND_REPORT"]
    assert rc == 0, "%s %s -> %d\n%s" % (entry["program"], " ".join(args), rc, out[-3000:])
    share = _device_share(out)
    assert share is not None, out[-1500:]
    pu_d, pu_r, pdd_d, pdd_r, pde_d, pde_r, ul_d, ul_r = share
    a = entry["args"]
    if entry["program"] == "pusch_test":
        uci = any(x in a for x in ("uci_ack", "cqi", "uci_ri"))
        assert (pu_r > 0 and pu_d == 0) if uci else (pu_d > 0 and pu_r == 0 and ul_d > 0 and ul_r == 0), (a, share)
    if entry["program"] == "phy_dl_test":
        tm = int(a[a.index("-t") + 1])
        assert pde_d > 0 and pde_r == 0, (a, share)  # every transport block is encoded on the device, whatever the transmission mode
        assert (pdd_d > 0 and pdd_r == 0) if tm == 1 else (pdd_r > 0), (a, share)  # one port, one antenna: decoded in one call too


# ---- the LUT rate (de)matchers (tests/ref_link/Makefile target `rmlut`): rm_turbo.c stays in a reference build whole (its non-LUT functions serve the
# sidelink / NB-IoT channels), and a definition in the program wins over the shared library's -- so the programs above run the REFERENCE's
# srsran_rm_turbo_{rx,tx}_lut.  Here those entry points are renamed away in the unmodified rm_turbo.o: sch.c / pssch.c call the library's per code block, and
# srsran_sch_init's srsran_rm_turbo_gentables() is the library's warm-start hook.  rm_turbo_test compares the library's LUT functions with the reference's own
# bit-by-bit srsran_rm_turbo_tx / _rx.
RMLUT_PROGRAMS = {"rm_turbo_test", "pdsch_test", "pusch_test", "pmch_test", "pssch_test"}


def _selected_rmlut():
    full = os.environ.get("REF_CTEST_FULL", "0") == "1"
    seen = {}
    for e in MANIFEST:
        if e["program"] not in RMLUT_PROGRAMS:
            continue
        n = seen.get(e["program"], 0)
        seen[e["program"]] = n + 1
        if full or e["program"] in ("rm_turbo_test", "pusch_test", "pmch_test") or n % 8 == 0:
            yield e


@pytest.mark.parametrize("entry", list(_selected_rmlut()), ids=lambda e: "rmlut:%s:%s" % (e["program"], e["name"]))
def test_ctest_line_on_the_librarys_lut_rate_matchers(entry, data_dir):
    import subprocess

    exe = os.path.join(HERE, "ref_link", "_build", "bin_rmlut", entry["program"])
    syms = subprocess.run(["nm", "-D", "--undefined-only", exe], stdout=subprocess.PIPE, text=True).stdout
    assert "srsran_rm_turbo_gentables" in syms, "bin_rmlut/%s does not take its LUT rate matchers from the library" % entry["program"]
    args = [str(data_dir / a[1:]) if a.startswith("@") else a for a in entry["args"]]
    rc, out = run_program("bin_rmlut", entry["program"], args, data_dir, timeout=900)
    assert rc == 0, "%s %s -> %d\n%s" % (entry["program"], " ".join(args), rc, out[-3000:])
