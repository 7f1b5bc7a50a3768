"""The reference's own test programs, unmodified, on top of the CPU ORACLE (tests/ref_link/oracle_shim.c puts liboracle.so behind the
reference entry points those programs call).  Runs without a GPU.

This pins the oracle's OFDM / DFT restatement (orc_ofdm.c, orc_dft_c) -- which has no compiled reference to be compared with, FFTW3 being
absent -- to the known answers the reference holds for srsran_ofdm_rx_sf: the reference's channel estimator and PBCH / PCFICH / PHICH /
PDCCH / PDSCH / PMCH decoders consume what orc_ofdm_rx_sf produces from the recorded captures and must decode the MIB of signal.1.92M.dat
(pbch_file_test.c:45-46,226-232), CFI 2 of signal.10M.dat (pcfich_file_test.c:251-255), the DCI and the PDSCH transport block of
signal.1.92M.amar.dat (pdcch_file_test.c:267-272, pdsch_pdcch_file_test.c:218), the PMCH transport block of the 100-PRB MBSFN subframe
(pmch_file_test.c:216-231); ofdm_test judges the modulator / demodulator pair by its loop-back criterion for 6 ... 110 PRB.  The turbo and LDPC
oracles, already pinned to oracle/_ref bit for bit, pass the reference's decoder tests here as well.

tests/test_gpu_ref_link.py runs the same programs on top of libsrsran_phy_hip.so on the GPU box.
"""
import re

import pytest
from ref_link_common import make_data_dir, run_program


@pytest.fixture(scope="module")
def data_dir(tmp_path_factory):
    return make_data_dir(tmp_path_factory.mktemp("ref_link_oracle_data"))


def _run(name, args, cwd, timeout=600):
    return run_program("bin_oracle", name, args, cwd, timeout)


@pytest.mark.parametrize("args", ["-r 1", "-e -r 1", "-s 0.5 -r 1", "-o 0.5 -r 1", "-N 4096 -r 1", "-e -o 0.5 -s 0.5 -N 4096 -r 1"])
def test_ofdm_test(args, data_dir):
    """dft/test/CMakeLists.txt:28-33"""
    rc, out = _run("ofdm_test", args.split(), data_dir)
    mse = [float(m) for m in re.findall(r"MSE=([0-9.]+)", out)]
    assert rc == 0 and len(mse) == 105 and max(mse) < 1e-4, out[-2000:]


def test_pbch_file_test(data_dir):
    rc, out = _run("pbch_file_test", ["-i", data_dir / "signal.1.92M.dat"], data_dir)
    assert rc == 0 and "This is the signal.1.92M.dat file" in out, out[-2000:]


def test_pcfich_file_test(data_dir):
    rc, out = _run("pcfich_file_test", ["-c", 150, "-n", 50, "-p", 2, "-i", data_dir / "signal.10M.dat"], data_dir)
    m = re.search(r"cfi: (\d+), distance: ([0-9.]+)", out)
    assert rc == 0 and m and int(m.group(1)) == 2 and float(m.group(2)) > 2.8, out[-2000:]


def test_phich_file_test(data_dir):
    rc, out = _run("phich_file_test", ["-c", 150, "-n", 50, "-p", 2, "-i", data_dir / "signal.10M.dat"], data_dir)
    assert rc == 0, out[-2000:]


def test_pdcch_file_test(data_dir):
    rc, out = _run("pdcch_file_test", ["-c", 1, "-f", 3, "-n", 6, "-p", 1, "-i", data_dir / "signal.1.92M.amar.dat"], data_dir)
    assert rc == 0 and "This is the file signal.1.92M.amar.dat" in out, out[-2000:]


def test_pdsch_pdcch_file_test(data_dir):
    rc, out = _run("pdsch_pdcch_file_test", ["-c", 1, "-f", 3, "-n", 6, "-p", 1, "-i", data_dir / "signal.1.92M.amar.dat"], data_dir)
    assert rc == 0 and "PDSCH Decoded OK!" in out, out[-2000:]


def test_pmch_file_test(data_dir):
    rc, out = _run("pmch_file_test", ["-i", data_dir / "pmch_100prbs_MCS2_SR0.bin"], data_dir)
    assert rc == 0 and "PMCH Decoded OK!" in out, out[-2000:]


@pytest.mark.parametrize("args", ["-n 100 -s 1 -l 504 -e 1.0 -t", "-n 20 -s 1 -l 6144 -e 1.5 -t", "-n 1 -s 1 -k -e 0.5"])
def test_turbodecoder_test(args, data_dir):
    """fec/turbo/test/CMakeLists.txt:45-48 (fewer frames for K = 6144: the scalar oracle takes 2 ms per block)"""
    rc, out = _run("turbodecoder_test", args.split(), data_dir)
    assert rc == 0 and "Done" in out, out[-2000:]


@pytest.mark.parametrize("length", [504, 6144])
def test_turbodecoder_test_error_free_at_high_snr(length, data_dir):
    rc, out = _run("turbodecoder_test", ("-n 10 -s 1 -l %d -e 6.0" % length).split(), data_dir)
    assert rc == 0 and "Done" in out and "Errors" not in out and re.search(r"10/10\s+BER: 0\.00e\+00", out), out[-2000:]


@pytest.mark.parametrize("bg,Z,sched", [(1, 2, 0), (1, 36, 0), (1, 384, 0), (2, 9, 0), (2, 208, 0), (2, 384, 0), (1, 384, 1)])
def test_ldpc_dec_c_test(bg, Z, sched, data_dir):
    """golden message / code-word pairs of examplesBG{1,2}.dat, exact (ldpc_dec_c_test.c:224-229)"""
    rc, out = _run("ldpc_dec_c_test", ["-b%d" % bg, "-l%d" % Z, "-x%d" % sched], data_dir)
    assert rc == 0 and "Test completed successfully" in out, out[-2000:]
