"""CPU-only checks of the product's host layer: the library loads, exports every symbol the public
headers declare, the struct layouts are those of the reference, and the host-side tables (QPP, base
graph, sizing helpers) equal the oracle's.  No kernel is launched (there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_api as O

ROOT = O.ROOT
INC = os.path.join(ROOT, "include", "srsran_amd")


@pytest.fixture(scope="module")
def L():
    from srslte_amd import build, capi

    build.build(verbose=False)
    return capi.lib()


def test_exports_every_declared_symbol(L):
    declared = set()
    for hdr in os.listdir(INC):
        txt = open(os.path.join(INC, hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        for m in re.finditer(r"SRSRAN_API[^;{]*?\b(\w+)\s*\(", txt):
            declared.add(m.group(1))
        for m in re.finditer(r"SRSRAN_API extern [^;]*?\b(\w+)\s*\[", txt):
            declared.add(m.group(1))
    declared -= {"SRSRAN_API", "__attribute__"}
    assert len(declared) > 60
    out = subprocess.check_output(["nm", "-D", "--defined-only", L._name], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    missing = sorted(declared - exported)
    assert not missing, "declared in include/ but not exported: %s" % missing


def test_struct_layout_matches_ctypes_mirror(L):
    """the C compiler's view of include/srsran_amd/phy_abi.h must equal the ctypes mirror (and, in the dev
    container, the reference headers: see test_struct_layout_matches_reference)"""
    from srslte_amd import capi

    sizes = _c_sizes(["-I", os.path.join(ROOT, "include")], OUR_INC)
    assert sizes["srsran_dft_plan_t"] == C.sizeof(capi.DftPlan)
    assert sizes["srsran_pss_t"] == C.sizeof(capi.Pss)
    assert sizes["srsran_sss_t"] == C.sizeof(capi.Sss)
    assert sizes["srsran_dft_precoding_t"] == C.sizeof(capi.DftPrecoding)
    assert sizes["off_pss_conv_output_avg"] == capi.Pss.conv_output_avg.offset
    assert sizes["srsran_ofdm_cfg_t"] == C.sizeof(capi.OfdmCfg)
    assert sizes["srsran_ofdm_t"] == C.sizeof(capi.Ofdm)
    assert sizes["srsran_tdec_t"] == C.sizeof(capi.Tdec)
    assert sizes["srsran_ldpc_decoder_t"] == C.sizeof(capi.LdpcDecoder)
    assert sizes["srsran_crc_t"] == C.sizeof(capi.Crc)
    assert sizes["srsran_sync_t"] == C.sizeof(capi.Sync)
    assert sizes["srsran_cfo_t"] == C.sizeof(capi.Cfo)
    assert sizes["srsran_cp_synch_t"] == C.sizeof(capi.CpSynch)
    assert sizes["off_sync_cfo_corr_frame"] == capi.Sync.cfo_corr_frame.offset
    assert sizes["off_sync_sss_signal"] == capi.Sync.sss_signal.offset
    assert sizes["srsran_ldpc_rm_t"] == C.sizeof(capi.LdpcRm) and sizes["off_ldpc_rm_Ncb"] == capi.LdpcRm.Ncb.offset
    assert sizes["srsran_ldpc_encoder_t"] == C.sizeof(capi.LdpcEncoder) and sizes["off_ldpc_encoder_encode"] == capi.LdpcEncoder.encode.offset


# sizeof/offsetof recorded from the reference headers (lib/include/srsran/phy/...) with gcc 11, x86-64
REFERENCE_LAYOUT = {"srsran_dft_plan_t": 48, "srsran_ofdm_cfg_t": 56, "srsran_ofdm_t": 272, "srsran_tdec_t": 18264,
                    "srsran_ldpc_decoder_t": 88, "srsran_crc_t": 2088, "srsran_tc_interl_t": 24,
                    "off_ofdm_tmp": 224, "off_tdec_interleaver": 208, "off_tdec_n_iter": 18256, "off_ldpc_decode_c": 80,
                    "srsran_pss_t": 35248, "srsran_sss_t": 38888, "srsran_dft_precoding_t": 5336,
                    "off_pss_conv_output_avg": 1864, "off_pss_tmp_ce": 34744, "off_sss_fc_tables": 3672,
                    "srsran_sync_t": 226864, "srsran_cfo_t": 40, "srsran_cp_synch_t": 16, "off_sync_cfo_corr_frame": 144800,
                    "off_sync_sss_signal": 194096, "srsran_ldpc_rm_t": 48, "srsran_ldpc_encoder_t": 80, "off_ldpc_rm_Ncb": 40,
                    "off_ldpc_encoder_encode": 48,
                    # the transport-block seam decode_tb_cb (sch.c:370): the soft buffer object and the head of srsran_sch_t
                    "srsran_softbuffer_rx_t": 40, "off_softbuffer_cb_crc": 24, "off_softbuffer_tb_crc": 32,
                    "off_sch_max_iterations": 0, "off_sch_avg_iterations": 4, "off_sch_llr_is_8bit": 8}


OUR_INC = '#define SCH_T srsran_hip_sch_head_t\n#include "srsran_amd/phy_abi.h"\n#include "srsran_amd/phy_sync_abi.h"\n#include "srsran_amd/phy_sch_abi.h"\n#include "srsran_amd/phy_modem_abi.h"\n#include "srsran_amd/phy_nr_sch_abi.h"\n'


def _c_sizes(flags, include, extra=""):
    src = include + """
#include <stdio.h>
#include <stddef.h>
int main(void) {
  printf("srsran_dft_plan_t %zu\\n", sizeof(srsran_dft_plan_t));
  printf("srsran_ofdm_cfg_t %zu\\n", sizeof(srsran_ofdm_cfg_t));
  printf("srsran_ofdm_t %zu\\n", sizeof(srsran_ofdm_t));
  printf("srsran_tdec_t %zu\\n", sizeof(srsran_tdec_t));
  printf("srsran_ldpc_decoder_t %zu\\n", sizeof(srsran_ldpc_decoder_t));
  printf("srsran_crc_t %zu\\n", sizeof(srsran_crc_t));
  printf("srsran_tc_interl_t %zu\\n", sizeof(srsran_tc_interl_t));
  printf("srsran_pss_t %zu\\n", sizeof(srsran_pss_t));
  printf("srsran_sss_t %zu\\n", sizeof(srsran_sss_t));
  printf("srsran_dft_precoding_t %zu\\n", sizeof(srsran_dft_precoding_t));
  printf("srsran_sync_t %zu\\n", sizeof(srsran_sync_t));
  printf("srsran_cfo_t %zu\\n", sizeof(srsran_cfo_t));
  printf("srsran_cp_synch_t %zu\\n", sizeof(srsran_cp_synch_t));
  printf("srsran_ldpc_rm_t %zu\\n", sizeof(srsran_ldpc_rm_t));
  printf("srsran_ldpc_encoder_t %zu\\n", sizeof(srsran_ldpc_encoder_t));
  printf("off_ldpc_rm_Ncb %zu\\n", offsetof(srsran_ldpc_rm_t, Ncb));
  printf("off_ldpc_encoder_encode %zu\\n", offsetof(srsran_ldpc_encoder_t, encode));
  printf("off_sync_cfo_corr_frame %zu\\n", offsetof(srsran_sync_t, cfo_corr_frame));
  printf("off_sync_sss_signal %zu\\n", offsetof(srsran_sync_t, sss_signal));
  printf("off_pss_conv_output_avg %zu\\n", offsetof(srsran_pss_t, conv_output_avg));
  printf("off_pss_tmp_ce %zu\\n", offsetof(srsran_pss_t, tmp_ce));
  printf("off_sss_fc_tables %zu\\n", offsetof(srsran_sss_t, fc_tables));
  printf("off_ofdm_tmp %zu\\n", offsetof(srsran_ofdm_t, tmp));
  printf("off_tdec_interleaver %zu\\n", offsetof(srsran_tdec_t, interleaver));
  printf("off_tdec_n_iter %zu\\n", offsetof(srsran_tdec_t, n_iter));
  printf("off_ldpc_decode_c %zu\\n", offsetof(srsran_ldpc_decoder_t, decode_c));
  printf("srsran_softbuffer_rx_t %zu\\n", sizeof(srsran_softbuffer_rx_t));
  printf("off_softbuffer_cb_crc %zu\\n", offsetof(srsran_softbuffer_rx_t, cb_crc));
  printf("off_softbuffer_tb_crc %zu\\n", offsetof(srsran_softbuffer_rx_t, tb_crc));
  printf("off_sch_max_iterations %zu\\n", offsetof(SCH_T, max_iterations));
  printf("off_sch_avg_iterations %zu\\n", offsetof(SCH_T, avg_iterations));
  printf("off_sch_llr_is_8bit %zu\\n", offsetof(SCH_T, llr_is_8bit));
  return 0; }
"""
    d = os.path.join(ROOT, "build", "scratch")
    os.makedirs(d, exist_ok=True)
    cfile, exe = os.path.join(d, "layout.c"), os.path.join(d, "layout")
    open(cfile, "w").write(src)
    subprocess.check_call(["gcc", "-std=gnu99"] + flags + [cfile, "-o", exe])
    return {k: int(v) for k, v in (ln.split() for ln in subprocess.check_output([exe], text=True).splitlines())}


def test_struct_layout_matches_recorded_reference(L):
    ours = _c_sizes(["-I", os.path.join(ROOT, "include")], OUR_INC)
    assert ours == REFERENCE_LAYOUT


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib/include"), reason="reference headers only exist in the dev container")
def test_struct_layout_matches_reference():
    inc = ('#define SCH_T srsran_sch_t\n#include <complex.h>\n#include "srsran/phy/phch/sch.h"\n#include "srsran/phy/dft/ofdm.h"\n#include "srsran/phy/fec/turbo/turbodecoder.h"\n'
           '#include "srsran/phy/fec/ldpc/ldpc_decoder.h"\n#include "srsran/phy/fec/crc.h"\n#include "srsran/phy/sync/pss.h"\n'
           '#include "srsran/phy/sync/sss.h"\n#include "srsran/phy/dft/dft_precoding.h"\n#include "srsran/phy/sync/sync.h"\n'
           '#include "srsran/phy/fec/ldpc/ldpc_rm.h"\n#include "srsran/phy/fec/ldpc/ldpc_encoder.h"\n')
    theirs = _c_sizes(["-I", "/root/reference/lib/include"], inc)
    assert theirs == REFERENCE_LAYOUT


def test_host_tables_match_oracle(L):
    from srslte_amd import capi

    # K table + QPP (natural and lane-order tables) for every size
    for idx, K in enumerate(O.tc_sizes()):
        assert L.srsran_cbsegm_cbsize(idx) == K and L.srsran_cbsegm_cbindex(K) == idx
        assert L.srsran_tdec_autoimp_get_subblocks(K) == O.orc().orc_tdec_autoimp_subblocks(K)
        assert L.srsran_tdec_autoimp_get_subblocks_8bit(K) == O.orc().orc_tdec_autoimp_subblocks_8bit(K)
        for win in (1, 8, 16):
            if K % win:
                continue
            it = capi.TcInterl()
            assert L.srsran_tc_interl_init(C.byref(it), K) == 0
            assert L.srsran_tc_interl_LTE_gen_interl(C.byref(it), K, win) == 0
            f, r = np.zeros(K, np.uint16), np.zeros(K, np.uint16)
            O.orc().orc_qpp_gen(K, win, O.P(f), O.P(r))
            assert np.array_equal(np.ctypeslib.as_array(it.forward, shape=(K,)), f)
            assert np.array_equal(np.ctypeslib.as_array(it.reverse, shape=(K,)), r)
            L.srsran_tc_interl_free(C.byref(it))
    assert L.srsran_cbsegm_cbindex(6145) == -1 and L.srsran_cbsegm_cbsize(188) == -1 and L.srsran_cbsegm_cbindex(41) == 1
    # LSindex + compact PCM
    ls = (C.c_uint8 * 385).in_dll(L, "LSindex")
    for z in range(385):
        want = O.orc().orc_ldpc_ls_index(z)
        assert ls[z] == (255 if want < 0 else want)
    pcm = np.zeros(46 * 68, np.uint16)
    assert L.create_compact_pcm(O.P(pcm), None, 0, 17) == -1
    # sizing helpers
    for prb in range(0, 113):
        assert L.srsran_symbol_sz(prb) == O.orc().orc_symbol_sz(prb)
        if prb:
            assert L.srsran_symbol_sz_power2(prb) == O.orc().orc_symbol_sz_power2(prb)


def test_rate_matching_tables_match_oracle(L):
    """the position tables (derived from TS 36.212 5.1.4.1 in the product) equal the restated reference builders
    (rm_turbo.c:175-273) for every block size, redundancy version and receiver layout"""
    orc = O.orc()
    orc.orc_rm_turbo_deinter.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    for K in O.tc_sizes():
        for rv in range(4):
            for nsb in (0, 8, 16, 32):
                if nsb and K % nsb:
                    assert L.srsran_hip_rm_turbo_table(O.P(np.zeros(3 * K + 12, np.uint16)), K, rv, nsb) == -2
                    continue
                a, b = np.zeros(3 * K + 12, np.uint16), np.zeros(3 * K + 12, np.uint16)
                assert L.srsran_hip_rm_turbo_table(O.P(a), K, rv, nsb) == 0
                assert orc.orc_rm_turbo_deinter(O.P(b), K, rv, nsb) == 0
                assert np.array_equal(a, b), (K, rv, nsb)
    assert L.srsran_hip_rm_turbo_table(O.P(a), 41, 0, 0) == -2 and L.srsran_hip_rm_turbo_table(O.P(a), 40, 4, 0) == -2


def test_cbsegm_matches_oracle_and_reference(L):
    """srsran_cbsegm (cbsegm.c:62-117): product, oracle and (dev container) the compiled reference agree"""
    from srslte_amd import capi

    ref = C.CDLL(O.REF_LIB) if (O.have_ref() and os.path.isdir("/root/reference")) else None
    rng = np.random.default_rng(2)
    for tbs in [0, 16, 40, 6120, 6128, 6144, 12216, 12240, 75376, 97896, 149776] + [int(8 * v) for v in rng.integers(1, 18000, 300)]:
        s = capi.Cbsegm()
        assert L.srsran_cbsegm(C.byref(s), tbs) == 0
        o = O.cbsegm(tbs)
        assert (s.C, s.K1, s.K2, s.C1, s.C2, s.F) == (o["C"], o["K1"], o["K2"], o["C1"], o["C2"], o["F"]), tbs
        if ref is not None:
            r = capi.Cbsegm()
            assert ref.srsran_cbsegm(C.byref(r), tbs) == 0
            assert [getattr(s, f) for f, _ in capi.Cbsegm._fields_[:11]] == [getattr(r, f) for f, _ in capi.Cbsegm._fields_[:11]], tbs


def test_no_gpu_means_loud_failure(L):
    """the product path must not fall back to the CPU: without a device every create/init fails"""
    from srslte_amd import capi

    if L.srsran_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert L.srsran_hip_tdec_batch_create(C.byref(h), 6144, 8, capi.TDEC_AUTO) == capi.SRSRAN_ERROR
    assert b"no HIP device" in L.srsran_hip_last_error()
    assert L.srsran_hip_ldpc_batch_create(C.byref(h), 0, 384, 0.8, 20, 1) == capi.SRSRAN_ERROR
    cfg = capi.OfdmCfg()
    cfg.nof_prb = 6
    assert L.srsran_hip_ofdm_batch_create(C.byref(h), C.byref(cfg), capi.DFT_FORWARD) == capi.SRSRAN_ERROR
    t = capi.Tdec()
    assert L.srsran_tdec_init(C.byref(t), 6144) == capi.SRSRAN_ERROR


def test_device_binding_bookkeeping_without_a_device(L):
    """srsran_hip_set_device / srsran_hip_set_thread_device on a box without a GPU: no device to bind to -> an error and a message, never a crash;
    returning a thread to the process default always works; the grant-level entry points and the warm-up fail loudly too"""
    from srslte_amd import capi

    if L.srsran_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    capi.lib()
    assert L.srsran_hip_set_device(0) == capi.SRSRAN_ERROR and b"no device 0" in L.srsran_hip_last_error()
    assert L.srsran_hip_set_thread_device(0) == capi.SRSRAN_ERROR and b"no device 0" in L.srsran_hip_last_error()
    assert L.srsran_hip_set_thread_device(3) == capi.SRSRAN_ERROR
    assert L.srsran_hip_set_thread_device(-1) == capi.SRSRAN_SUCCESS
    assert L.srsran_hip_get_thread_device() == 0
    assert L.srsran_hip_warmup(2) == capi.SRSRAN_ERROR
    L.srsran_rm_turbo_gentables()  # the init-time hook of srsran_sch_init: a no-op without a device
    g = capi.HipPdschRx(capi.HipGrantTb(2, 6200, 0, 2400, 1, 8, 0, 1), 1.0, 0.0)
    x = (C.c_float * 4800)()
    rows = (C.c_void_p * 2)()
    sb = capi.SoftbufferRx(2, 18600, rows, rows, None, False)
    res = capi.HipGrantRes()
    assert L.srsran_hip_pdsch_decode(C.byref(g), x, None, C.byref(sb), x, C.byref(res)) == capi.SRSRAN_ERROR


def test_product_never_touches_the_oracle():
    """nothing under srslte_amd/ or include/ may reference oracle/ (the oracle is the checker, not the product)"""
    bad = []
    for base in ("srslte_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"oracle[/_.]|liboracle|orc_", txt) and "oracle" in txt.replace("the oracle", ""):
                        if re.search(r'import oracle|oracle/lib|liboracle|#include "../oracle|orc_[a-z]+\(', txt):
                            bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_transport_block_seams_are_bound_in_the_reference_programs():
    """tests/ref_link `make tb`: in the reference's unmodified sch.o / sch_nr.o the seam symbols are WEAK definitions, and the linked programs call the
    library's entry points through the bindings (static facts of the build, no GPU needed; the programs themselves run in tests/test_gpu_ref_ctest.py)"""
    build = os.path.join(ROOT, "tests", "ref_link", "_build")
    weak = {"sch_weak.o": {"decode_tb_cb", "srsran_dlsch_encode2"},
            "sch_nr_weak.o": {"srsran_dlsch_nr_decode", "srsran_ulsch_nr_decode", "srsran_dlsch_nr_encode", "srsran_ulsch_nr_encode"}}
    if not os.path.exists(os.path.join(build, "bin_tb", "pusch_test")):
        pytest.skip("tests/ref_link/_build/bin_tb not built (needs the reference tree: dev container)")
    for obj, names in weak.items():
        p = os.path.join(build, "obj", "tb", obj)
        if os.path.exists(p):  # (objects stay in the dev container's tree; the programs travel)
            syms = dict((ln.split()[-1], ln.split()[-2]) for ln in subprocess.check_output(["nm", p], text=True).splitlines() if len(ln.split()) >= 3)
            assert all(syms.get(n) == "W" for n in names), (obj, {n: syms.get(n) for n in names})
    und = lambda prog: {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--undefined-only", os.path.join(build, "bin_tb", prog)], text=True).splitlines()}
    lte, nr = und("pusch_test"), und("pdsch_nr_test")
    assert {"srsran_hip_decode_tb_cb", "srsran_hip_encode_tb"} <= lte, sorted(n for n in lte if "hip" in n)
    assert {"srsran_hip_sch_nr_decode_tb", "srsran_hip_sch_nr_encode_tb"} <= nr, sorted(n for n in nr if "hip" in n)
    # the plain link set of the same program binds the per-code-block entry points instead
    full = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--undefined-only", os.path.join(build, "bin_full", "pusch_test")], text=True).splitlines()}
    assert "srsran_hip_decode_tb_cb" not in full and "srsran_tdec_iteration" in full
