#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (dev container only: needs /root/reference and
oracle/_ref/libsrsran_ref.so built by `make -C oracle ref`).

A fixture is DATA: inputs and the outputs the reference produced for them, plus data the reference's
own tests hold (the K=504 known-answer block of turbodecoder_test.h; rows of examplesBG{1,2}.dat).
No reference source text is stored.

  tests/golden/turbo_ref.npz   reference srsran_tdec_run_all outputs on seeded noisy LLRs; known-answer
                               K=504 message/code word; CRC32 of every QPP table the reference builds
  tests/golden/turbo8_ref.npz  reference 8-bit turbo decoders (sse8 / avx8 window) on seeded int8 LLRs
  tests/golden/syncglue_ref.npz  reference srsran_cfo_correct / srsran_cp_synch outputs on seeded inputs
  tests/golden/ldpc_ref.npz    reference srsran_ldpc_decoder_decode_c (scalar C and AVX2) outputs on seeded LLRs
  tests/golden/ldpc_fs_ref.npz reference float / int16 LDPC decoder outputs on seeded LLRs
  tests/golden/rm_ref.npz      reference srsran_rm_turbo_rx_lut_ / _8bit: seeded inputs, CRC32 of the resulting soft buffers
  tests/golden/modem_ref.npz   reference srsran_demod_soft_demodulate{,_s,_b} outputs on seeded symbols (all five
                               modulations, lengths around the SIMD group sizes, in- and out-of-range amplitudes);
                               scrambling chips recovered from srsran_sequence_apply_s / pusch / pdsch apply
  tests/golden/mod_ref.npz     reference srsran_mod_modulate_bytes outputs (every constellation point of the five tables; seeded bits behind srsran_sequence_apply_packed)
  tests/golden/ldpc_tx_ref.npz reference LDPC encoder (C and AVX2: equal) code words with and without filler bits, srsran_ldpc_rm_tx
                               outputs and srsran_ldpc_rm_rx_{c,s,f} soft buffers (as CRC32) on stored inputs
  tests/golden/sch_tx_ref.npz  transmit side of transport blocks as encode_tb_off (sch.c:238-345) chains the reference's
                               srsran_crc_*, srsran_tcod_encode_lut and srsran_rm_turbo_tx_lut: payload bytes -> packed e bits
  tests/golden/ldpc_flood_ref.npz reference SRSRAN_LDPC_DECODER_C_FLOOD (scalar flooded schedule) outputs on seeded LLRs
  tests/golden/tcod_lut_ref.npz reference srsran_tcod_encode_lut (inputs after the call, parity bytes, running CRC state) and
                               srsran_rm_turbo_tx_lut (outputs at several bit offsets / lengths / redundancy versions)
  tests/golden/sch_nr_ref.npz  NR transport blocks through the reference's blocks in the order of sch_nr.c (segmentation, CRCs, LDPC encoder,
                               rate matcher both ways, decoder with CRC early stop): LLRs in, verdicts / iterations / payload out
  tests/golden/ldpc_examples.npz  subset of the reference's golden message/code-word pairs
  tests/golden/ref_link_data.npz  DATA files the reference's own test programs read (tests/ref_link runs those programs, unmodified,
                               on top of the product library): phch/test/pmch_100prbs_MCS2_SR0.bin (pmch_file_test) and the ten
                               message / code-word pairs of fec/ldpc/test/examplesBG{1,2}.dat for a few lifting sizes (ldpc_dec_c_test)
  tests/golden/sync_captures.npz  the recorded air captures the reference's own tests hold for the PSS / SSS path
                               (phch/test/signal.1.92M.dat: pbch_file_test, cell 150; signal.1.92M.amar.dat: pdcch_file_test -c 1;
                               signal.10M.dat: pcfich_file_test -c 150 -n 50) with the cell ids those tests are given
"""
import ctypes as C
import os
import re
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api as O  # noqa: E402

REF = "/root/reference/lib"
OUT = os.path.join(ROOT, "tests", "golden")
ref = C.CDLL(O.REF_LIB)
P = O.P


class Interl(C.Structure):
    _fields_ = [("forward", C.POINTER(C.c_uint16)), ("reverse", C.POINTER(C.c_uint16)), ("max_long_cb", C.c_uint32)]


def c_array(text, name):
    m = re.search(name + r"\[[^\]]*\]\s*=\s*\{([^}]*)\}", text, re.S)
    return np.array([int(x) for x in re.findall(r"-?\d+", m.group(1))], dtype=np.uint8)


def turbo():
    d = {}
    txt = open(os.path.join(REF, "src/phy/fec/turbo/test/turbodecoder_test.h")).read()
    d["known_data"] = c_array(txt, "known_data")
    d["known_data_encoded"] = c_array(txt, "known_data_encoded")
    assert d["known_data"].size == 504 and d["known_data_encoded"].size == 3 * 504 + 12
    h = C.create_string_buffer(64 * 1024)
    assert ref.srsran_tdec_init(h, 6144) == 0
    ref.srsran_tdec_force_not_sb(h)
    cases = []
    for K, n_cb in ((40, 6), (504, 4), (1024, 4), (6144, 3)):
        for snr in (2.0, -1.0, -4.0):
            _, llr = O.turbo_llrs(K, n_cb, snr, seed=K * 3 + int(snr * 10))
            outs = np.zeros((8, n_cb, K // 8), np.uint8)
            for nit in range(1, 9):
                for i in range(n_cb):
                    assert ref.srsran_tdec_run_all(h, P(llr[i]), P(outs[nit - 1, i]), nit, K) == 0
            key = "K%d_snr%d" % (K, int(snr * 10))
            d[key + "_llr"] = llr
            d[key + "_out"] = outs
            cases.append(key)
    d["cases"] = np.array(cases)
    # QPP tables: crc32 of forward|reverse for every K and window 1/8/16 (window only where K % win == 0)
    crcs = []
    ref.srsran_cbsegm_cbsize.restype = C.c_int
    for idx in range(188):
        K = ref.srsran_cbsegm_cbsize(idx)
        for win in (1, 8, 16):
            if K % win:
                crcs.append((K, win, 0))
                continue
            it = Interl()
            assert ref.srsran_tc_interl_init(C.byref(it), K) == 0
            assert ref.srsran_tc_interl_LTE_gen_interl(C.byref(it), K, win) == 0
            f = np.ctypeslib.as_array(it.forward, shape=(K,)).copy()
            r = np.ctypeslib.as_array(it.reverse, shape=(K,)).copy()
            crcs.append((K, win, zlib.crc32(f.tobytes() + r.tobytes())))
            if (K, win) in ((40, 1), (6144, 16)):
                d["qpp_K%d_w%d_fwd" % (K, win)] = f
            ref.srsran_tc_interl_free(C.byref(it))
    d["qpp_crc"] = np.array(crcs, dtype=np.uint32)
    d["autoimp"] = np.array([[ref.srsran_cbsegm_cbsize(i), ref.srsran_tdec_autoimp_get_subblocks(ref.srsran_cbsegm_cbsize(i)),
                              ref.srsran_tdec_autoimp_get_subblocks_8bit(ref.srsran_cbsegm_cbsize(i))] for i in range(188)], dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "turbo_ref.npz"), **d)
    print("turbo_ref.npz", os.path.getsize(os.path.join(OUT, "turbo_ref.npz")))


def turbo8():
    """reference 8-bit decoders: AUTO through srsran_tdec_run_all_8bit; the manually selected sse8 / avx8 decoders
    through srsran_tdec_run_all on int8-valued int16 LLRs (the reference's 8-bit entry point cannot run them)"""
    d = {}
    cases = []
    for impl, sizes in ((0, ((40, 4), (504, 3), (816, 3), (1024, 3), (2112, 3), (6144, 3))), (6, ((816, 2), (6144, 2))),
                        (7, ((1344, 2), (6144, 2)))):
        h = C.create_string_buffer(64 * 1024)
        assert ref.srsran_tdec_init_manual(h, 6144, impl) == 0
        ref.srsran_tdec_force_not_sb(h)
        for K, n_cb in sizes:
            for snr, scale in ((2.0, 12.0), (-1.0, 12.0), (0.0, 60.0)):
                _, llr = O.turbo_llrs_8bit(K, n_cb, snr, seed=K * 3 + impl + int(scale), scale=scale)
                outs = np.zeros((8, n_cb, K // 8), np.uint8)
                for nit in range(1, 9):
                    for i in range(n_cb):
                        if impl == 0:
                            assert ref.srsran_tdec_run_all_8bit(h, P(llr[i].copy()), P(outs[nit - 1, i]), nit, K) == 0
                        else:
                            assert ref.srsran_tdec_run_all(h, P(llr[i].astype(np.int16)), P(outs[nit - 1, i]), nit, K) == 0
                key = "i%d_K%d_snr%d_s%d" % (impl, K, int(snr * 10), int(scale))
                d[key + "_llr"] = llr
                d[key + "_out"] = outs
                cases.append(key)
        ref.srsran_tdec_free(h)
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "turbo8_ref.npz"), **d)
    print("turbo8_ref.npz", os.path.getsize(os.path.join(OUT, "turbo8_ref.npz")))


def ldpc_fs():
    """reference float (type 0) and int16 (type 1) layered decoders on seeded LLRs"""
    d = {}
    cases = []
    for bg, Z in ((0, 384), (1, 384), (0, 2), (1, 9), (0, 112), (1, 208), (0, 24)):
        for snr, sf, nit in ((2.0, 0.8, 10), (0.0, 0.75, 4)):
            g = O.ldpc_graph(bg, Z)
            K, N = g.bgK * Z, g.bgN * Z
            _, l8 = O.ldpc_llrs(bg, Z, 2, snr, seed=Z + bg * 1000 + 5, clip=127)
            rng = np.random.default_rng(Z)
            rm = N - 2 * Z if nit == 10 else (g.bgK + 9) * Z + 3
            for typ, llrs in ((0, (l8 * rng.uniform(0.3, 0.35, l8.shape)).astype(np.float32)),
                              (1, (l8.astype(np.int32) * 180).clip(-32767, 32767).astype(np.int16))):
                dec = C.create_string_buffer(4096)
                a = Args(typ, bg, Z, sf, nit)
                assert ref.srsran_ldpc_decoder_init(dec, C.byref(a)) == 0
                o = np.zeros((2, K), np.uint8)
                fn = ref.srsran_ldpc_decoder_decode_f if typ == 0 else ref.srsran_ldpc_decoder_decode_s
                for i in range(2):
                    assert fn(dec, P(llrs[i]), P(o[i]), rm) == nit
                ref.srsran_ldpc_decoder_free(dec)
                key = "t%d_bg%d_z%d_it%d" % (typ, bg, Z, nit)
                d[key + "_llr"], d[key + "_out"] = llrs, np.packbits(o, axis=1)
                d[key + "_par"] = np.array([bg, Z, nit, rm, int(sf * 100)], dtype=np.int32)
                cases.append(key)
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "ldpc_fs_ref.npz"), **d)
    print("ldpc_fs_ref.npz", os.path.getsize(os.path.join(OUT, "ldpc_fs_ref.npz")))


def rm():
    """reference srsran_rm_turbo_rx_lut_ / _8bit outputs on seeded soft bits (soft buffer pre-loaded, as after an earlier HARQ round)"""
    ref.srsran_rm_turbo_gentables()
    ref.srsran_rm_turbo_rx_lut_.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_bool]
    rng = np.random.default_rng(21)
    sizes = O.tc_sizes()
    recs, ins, outs, bases = [], [], [], []
    for ci in (0, 40, 100, 125, 187):
        K = sizes[ci]
        for rv in (range(4) if K < 6144 else (0, 3)):
            for E in ((K // 2 + 5, 3 * K + 100, 7 * K + 33) if K < 2000 else (K + 7, 3 * K + 100)):
                for mode in (0, 1, 2):  # natural int16, decoder layout int16, int8
                    dt = np.int8 if mode == 2 else np.int16
                    amp = 60 if mode == 2 else 3000
                    x = rng.integers(-amp, amp, E).astype(dt)
                    base = rng.integers(-amp // 8, amp // 8, 3 * (K + 32) + 12).astype(dt)
                    o = base.copy()
                    if mode == 2:
                        assert ref.srsran_rm_turbo_rx_lut_8bit(P(x), P(o), E, ci, rv) == 0
                    else:
                        assert ref.srsran_rm_turbo_rx_lut_(P(x), P(o), E, ci, rv, bool(mode)) == 0
                    recs.append((K, rv, E, mode))
                    ins.append(x.astype(np.int16))
                    bases.append(base.astype(np.int16))
                    outs.append(zlib.crc32(o.tobytes()))
    d = {"recs": np.array(recs, np.int32), "in": np.concatenate(ins), "base": np.concatenate(bases), "out_crc": np.array(outs, np.uint32)}
    np.savez_compressed(os.path.join(OUT, "rm_ref.npz"), **d)
    print("rm_ref.npz", os.path.getsize(os.path.join(OUT, "rm_ref.npz")))


def syncglue():
    """reference srsran_cfo_correct and srsran_cp_synch (cfo.c, cp.c, cexptab.c need no FFT library) on seeded inputs"""
    d = {}
    rng = np.random.default_rng(77)
    n = 1920
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
    d["cfo_x"] = x
    freqs = np.array([0.0, 1.3e-4, -7.7e-4, 0.013, -0.2], np.float32)
    d["cfo_freqs"] = freqs
    outs = np.zeros((freqs.size, n), np.complex64)
    h = C.create_string_buffer(256)
    assert ref.srsran_cfo_init(h, n) == 0
    ref.srsran_cfo_correct.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
    for i, f in enumerate(freqs):
        ref.srsran_cfo_correct(h, P(x), P(outs[i]), float(f))
    ref.srsran_cfo_free(h)
    d["cfo_out"] = outs
    N, nsym, max_off = 512, 7, 200
    cp = 36
    y = ((rng.standard_normal(max_off + (nsym + 1) * (N + cp + 1) + N) + 1j * rng.standard_normal(max_off + (nsym + 1) * (N + cp + 1) + N)) * 0.7)
    y = y.astype(np.complex64)
    q = C.create_string_buffer(64)
    assert ref.srsran_cp_synch_init(q, N) == 0
    ref.srsran_cp_synch.restype = C.c_uint32
    idx = ref.srsran_cp_synch(q, P(y), max_off, nsym, cp)
    corr_ptr = C.cast(q, C.POINTER(C.c_void_p))[0]
    corr = np.ctypeslib.as_array(C.cast(corr_ptr, C.POINTER(C.c_float)), shape=(2 * max_off,)).copy().view(np.complex64)
    d["cp_y"], d["cp_par"], d["cp_idx"], d["cp_corr"] = y, np.array([N, nsym, max_off, cp]), np.array([idx]), corr
    ref.srsran_cp_synch_free(q)
    np.savez_compressed(os.path.join(OUT, "syncglue_ref.npz"), **d)
    print("syncglue_ref.npz", os.path.getsize(os.path.join(OUT, "syncglue_ref.npz")))


class Args(C.Structure):
    _fields_ = [("type", C.c_int), ("bg", C.c_int), ("ls", C.c_uint16), ("scaling_fctr", C.c_float), ("max_nof_iter", C.c_uint32)]


def sync_captures():
    """data files of the reference's tests + the answers its CMake test lines state (phch/test/CMakeLists.txt:433,439-442;
    pbch_file_test.c:32-36 default cell id 150).  Samples are stored as they are (complex float32)."""
    d = {}
    base = os.path.join(REF, "src/phy/phch/test")
    # name, file, fft size the capture's sampling rate corresponds to, samples used, cell id of the test line
    for name, fn, N, n_use, cell in (("pbch_1_92M", "signal.1.92M.dat", 128, 9600, 150),
                                     ("amar_1_92M_sf0", "signal.1.92M.amar.dat", 128, 9600, 1),
                                     ("pcfich_10M", "signal.10M.dat", 1024, 7680, 150)):
        x = np.fromfile(os.path.join(base, fn), dtype=np.complex64)
        d[name + "_x"] = x
        d[name + "_par"] = np.array([N, n_use, cell], dtype=np.int32)
    d["cases"] = np.array(["pbch_1_92M", "amar_1_92M_sf0", "pcfich_10M"])
    np.savez_compressed(os.path.join(OUT, "sync_captures.npz"), **d)
    print("sync_captures.npz", os.path.getsize(os.path.join(OUT, "sync_captures.npz")))


def ref_link():
    """inputs of the reference's test programs that tests/ref_link re-materialises on the GPU box (data only)."""
    d = {}
    d["pmch_100prb_x"] = np.fromfile(os.path.join(REF, "src/phy/phch/test/pmch_100prbs_MCS2_SR0.bin"), dtype=np.complex64)
    conv = lambda s: np.array([2 if c == "-" else int(c) for c in s], dtype=np.uint8)
    for bg, sizes in ((0, (2, 36, 208, 384)), (1, (9, 15, 208, 384))):
        lines = open(os.path.join(REF, "src/phy/fec/ldpc/test/examplesBG%d.dat" % (bg + 1))).read().split("\n")
        sect, cur = {}, None
        for ln in lines:
            if ln.startswith("ls"):
                cur = ln.strip()
                sect[cur] = []
            elif ln.strip() and cur:
                sect[cur].append(ln.strip())
        for Z in sizes:
            for part in ("msgs", "cwds"):
                rows = np.stack([conv(x) for x in sect["ls%d%s" % (Z, part)]])
                assert rows.shape[0] == 10
                # 0 / 1 / filler ('-') -> two bit planes
                d["bg%d_z%d_%s" % (bg, Z, part)] = np.packbits(rows == 1, axis=1)
                d["bg%d_z%d_%s_fill" % (bg, Z, part)] = np.packbits(rows == 2, axis=1)
                d["bg%d_z%d_%s_len" % (bg, Z, part)] = np.array([rows.shape[1]], dtype=np.int32)
    d["ldpc_sizes"] = np.array([[0, 2], [0, 36], [0, 208], [0, 384], [1, 9], [1, 15], [1, 208], [1, 384]], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "ref_link_data.npz"), **d)
    print("ref_link_data.npz", os.path.getsize(os.path.join(OUT, "ref_link_data.npz")))


def ldpc():
    d = {}
    cases = []
    for bg, Z in ((0, 384), (1, 384), (0, 2), (1, 9), (0, 112), (1, 208), (0, 15), (0, 24)):
        for snr, sf, nit in ((2.0, 0.8, 20), (0.0, 0.75, 5)):
            g = O.ldpc_graph(bg, Z)
            K, N = g.bgK * Z, g.bgN * Z
            _, llrs = O.ldpc_llrs(bg, Z, 2, snr, seed=Z + bg * 1000, clip=63 if snr > 1 else 127)
            rm = N - 2 * Z if nit == 20 else (g.bgK + 9) * Z + 3
            outs = {}
            for typ in (2, 4):
                dec = C.create_string_buffer(4096)
                a = Args(typ, bg, Z, sf, nit)
                assert ref.srsran_ldpc_decoder_init(dec, C.byref(a)) == 0
                o = np.zeros((2, K), np.uint8)
                for i in range(2):
                    assert ref.srsran_ldpc_decoder_decode_c(dec, P(llrs[i]), P(o[i]), rm) == nit
                ref.srsran_ldpc_decoder_free(dec)
                outs[typ] = o
            assert np.array_equal(outs[2], outs[4])
            key = "bg%d_z%d_it%d" % (bg, Z, nit)
            d[key + "_llr"], d[key + "_out"] = llrs, np.packbits(outs[2], axis=1)
            d[key + "_par"] = np.array([bg, Z, nit, rm, int(sf * 100)], dtype=np.int32)
            cases.append(key)
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "ldpc_ref.npz"), **d)
    print("ldpc_ref.npz", os.path.getsize(os.path.join(OUT, "ldpc_ref.npz")))
    # golden examples of the reference's own tests
    ex = {}
    for bg in (0, 1):
        lines = open(os.path.join(REF, "src/phy/fec/ldpc/test/examplesBG%d.dat" % (bg + 1))).read().split("\n")
        sect = {}
        cur = None
        for ln in lines:
            if ln.startswith("ls"):
                cur = ln.strip()
                sect[cur] = []
            elif ln.strip() and cur:
                sect[cur].append(ln.strip())
        for Z in (2, 3, 5, 7, 9, 11, 13, 15, 16, 36, 104, 208, 384):
            conv = lambda s: np.array([254 if c == "-" else int(c) for c in s], dtype=np.uint8)
            ex["bg%d_z%d_msgs" % (bg, Z)] = np.packbits(np.stack([conv(s) for s in sect["ls%dmsgs" % Z][:3]]) == 1, axis=1)
            ex["bg%d_z%d_fill" % (bg, Z)] = np.packbits(np.stack([conv(s) for s in sect["ls%dmsgs" % Z][:3]]) == 254, axis=1)
            ex["bg%d_z%d_cwds" % (bg, Z)] = np.packbits(np.stack([conv(s) for s in sect["ls%dcwds" % Z][:3]]) == 1, axis=1)
            ex["bg%d_z%d_cfill" % (bg, Z)] = np.packbits(np.stack([conv(s) for s in sect["ls%dcwds" % Z][:3]]) == 254, axis=1)
    np.savez_compressed(os.path.join(OUT, "ldpc_examples.npz"), **ex)
    print("ldpc_examples.npz", os.path.getsize(os.path.join(OUT, "ldpc_examples.npz")))


def ldpc_flood():
    class Args(C.Structure):
        _fields_ = [("type", C.c_int), ("bg", C.c_int), ("ls", C.c_uint16), ("scaling_fctr", C.c_float), ("max_nof_iter", C.c_int)]

    d, cases = {}, []
    for bg, Z, snr, nit, sf in ((0, 384, 2.0, 5, 0.8), (0, 384, 0.5, 10, 0.75), (1, 208, 1.0, 6, 0.8), (0, 36, 3.0, 4, 1.0), (1, 7, 2.0, 8, 0.6),
                                (0, 2, 4.0, 10, 0.8), (1, 128, 0.0, 6, 0.8)):
        g = O.ldpc_graph(bg, Z)
        K, N = g.bgK * Z, g.bgN * Z
        for clip in (63, 127):
            _, llrs = O.ldpc_llrs(bg, Z, 2, snr, seed=Z + bg, clip=clip)
            for rm in (N - 2 * Z, (g.bgK + 9) * Z + 3):
                dec = C.create_string_buffer(4096)
                a = Args(3, bg, Z, sf, nit)  # SRSRAN_LDPC_DECODER_C_FLOOD
                assert ref.srsran_ldpc_decoder_init(dec, C.byref(a)) == 0
                o = np.zeros((2, K), np.uint8)
                for i in range(2):
                    assert ref.srsran_ldpc_decoder_decode_c(dec, P(llrs[i]), P(o[i]), rm) == nit
                ref.srsran_ldpc_decoder_free(dec)
                key = "bg%d_z%d_it%d_c%d_rm%d" % (bg, Z, nit, clip, rm)
                d[key + "_llr"], d[key + "_out"] = llrs, np.packbits(o, axis=1)
                d[key + "_par"] = np.array([bg, Z, nit, rm, int(round(sf * 100))], dtype=np.int32)
                cases.append(key)
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "ldpc_flood_ref.npz"), **d)
    print("ldpc_flood_ref.npz", os.path.getsize(os.path.join(OUT, "ldpc_flood_ref.npz")))


def ldpc_tx():
    d = {}
    rng = np.random.default_rng(77)
    enc_cases, rm_cases = [], []
    QMS = [1, 2, 4, 6, 8]
    for bg, ls in ((0, 384), (0, 2), (1, 208), (0, 52), (1, 11), (1, 384), (0, 224), (1, 15)):
        N, K = ls * (66 if bg == 0 else 50), ls * (22 if bg == 0 else 10)
        for F in (0, min(ls, 20)):
            msg = rng.integers(0, 2, K).astype(np.uint8)
            if F:
                msg[K - F:] = 254
            for rmlen in (N, N // 2 + 3):
                outs = []
                for typ in (0, 1):
                    enc = C.create_string_buffer(256)
                    assert ref.srsran_ldpc_encoder_init(enc, typ, bg, C.c_uint16(ls)) == 0
                    o = np.full(N, 7, np.uint8)
                    assert ref.srsran_ldpc_encoder_encode_rm(enc, P(msg), P(o), C.c_uint32(K), C.c_uint32(rmlen)) == 0
                    ref.srsran_ldpc_encoder_free(enc)
                    outs.append(o)
                assert np.array_equal(outs[0], outs[1])
                key = "enc_bg%d_z%d_f%d_r%d" % (bg, ls, F, rmlen)
                d[key + "_msg"], d[key + "_cw"] = msg, outs[0]
                enc_cases.append(key)
            cw = d["enc_bg%d_z%d_f%d_r%d_cw" % (bg, ls, F, N)]
            for rv, mod, nref_on, emul in ((0, 1, 0, 0.6), (1, 3, 0, 1.0), (2, 2, 1, 0.4), (3, 4, 0, 2.3), (0, 0, 1, 1.5))[:5 if ls < 200 else (2 if F else 0)]:
                Qm = QMS[mod]
                Nref = (N * 2 // 3 // ls * ls + 7) if nref_on else N
                E = max(Qm, int(N * emul) // Qm * Qm)
                q = C.create_string_buffer(64)
                assert ref.srsran_ldpc_rm_tx_init(q) == 0
                tx = np.zeros(E, np.uint8)
                ref.srsran_ldpc_rm_tx(q, P(cw), P(tx), C.c_uint32(E), C.c_int(bg), C.c_uint32(ls), C.c_uint8(rv), C.c_int(mod), C.c_uint32(Nref))
                ref.srsran_ldpc_rm_tx_free(q)
                key = "rm_bg%d_z%d_f%d_rv%d_m%d_n%d_e%d" % (bg, ls, F, rv, mod, Nref, E)
                d[key + "_tx"] = np.packbits(tx)
                x8 = rng.integers(-50, 51, E).astype(np.int8)
                base = rng.integers(-20, 21, N).astype(np.int8)
                d[key + "_x"], d[key + "_base"] = x8, base
                crcs = []
                for dt, init, fn, free, mul in ((np.int8, ref.srsran_ldpc_rm_rx_init_c, ref.srsran_ldpc_rm_rx_c, ref.srsran_ldpc_rm_rx_free_c, 1),
                                                (np.int16, ref.srsran_ldpc_rm_rx_init_s, ref.srsran_ldpc_rm_rx_s, ref.srsran_ldpc_rm_rx_free_s, 200),
                                                (np.float32, ref.srsran_ldpc_rm_rx_init_f, ref.srsran_ldpc_rm_rx_f, ref.srsran_ldpc_rm_rx_free_f, 0.25)):
                    q = C.create_string_buffer(64)
                    assert init(q) == 0
                    x = (x8.astype(np.float64) * mul).astype(dt)
                    o = (base.astype(np.float64) * mul).astype(dt)
                    r = fn(q, P(x), P(o), C.c_uint32(E), C.c_uint32(F), C.c_int(bg), C.c_uint32(ls), C.c_uint8(rv), C.c_int(mod), C.c_uint32(Nref))
                    free(q)
                    crcs += [zlib.crc32(o.tobytes()), r]
                d[key + "_rx"] = np.array(crcs, dtype=np.int64)
                rm_cases.append(key)
    d["enc_cases"], d["rm_cases"] = np.array(enc_cases), np.array(rm_cases)
    np.savez_compressed(os.path.join(OUT, "ldpc_tx_ref.npz"), **d)
    print("ldpc_tx_ref.npz", os.path.getsize(os.path.join(OUT, "ldpc_tx_ref.npz")))


def sch_tx():
    class Tcod(C.Structure):
        _fields_ = [("max_long_cb", C.c_uint32), ("temp", C.c_void_p)]

    class Seg(C.Structure):
        _fields_ = [(n, C.c_uint32) for n in ("F", "C", "K1", "K2", "K1_idx", "K2_idx", "C1", "C2", "tbs", "L_tb", "L_cb", "Z")]

    crc_tb, crc_cb = C.create_string_buffer(4096), C.create_string_buffer(4096)
    assert ref.srsran_crc_init(crc_tb, C.c_uint32(0x1864CFB), 24) == 0 and ref.srsran_crc_init(crc_cb, C.c_uint32(0x1800063), 24) == 0
    tc = Tcod()
    assert ref.srsran_tcod_init(C.byref(tc), 6144) == 0
    ref.srsran_rm_turbo_gentables()
    rng = np.random.default_rng(31)
    d, cases = {}, []
    for tbs, Qm, rv, G in ((40, 2, 0, 120), (6200, 6, 0, 9000), (75376, 6, 0, 100800), (75376, 6, 2, 100800), (12960, 4, 1, 21004),
                           (6264, 2, 3, 7000), (31704, 8, 0, 40000), (2216, 2, 2, 9000)):
        seg = Seg()
        assert ref.srsran_cbsegm(C.byref(seg), tbs) == 0 and seg.F == 0
        data = rng.integers(0, 256, tbs // 8 + 3).astype(np.uint8)
        data[tbs // 8:] = 0
        e_bits = np.zeros(G // 8 + 8, np.uint8)
        cb_in, parity = np.zeros(6144 // 8 + 64, np.uint8), np.zeros(3 * 6144 // 8 + 64, np.uint8)
        Gp = G // Qm
        gamma = Gp % seg.C
        ref.srsran_crc_set_init(crc_tb, C.c_uint64(0))
        rp = wp = 0
        for i in range(seg.C):  # sch.c:280-340
            cb_len, idx = (seg.K2, seg.K2_idx) if i < seg.C2 else (seg.K1, seg.K1_idx)
            rlen = cb_len - 24 if seg.C > 1 else cb_len
            n_e = Qm * (Gp // seg.C) if i <= seg.C - gamma - 1 else Qm * -(-Gp // seg.C)
            last = i == seg.C - 1
            nb = (rlen - 24) // 8 if last else rlen // 8
            cb_in[:] = 0
            cb_in[:nb] = data[rp // 8:rp // 8 + nb]
            ref.srsran_tcod_encode_lut(C.byref(tc), crc_tb, crc_cb if seg.C > 1 else None, P(cb_in), P(parity), C.c_uint32(idx), C.c_bool(last))
            w_buff = np.zeros(3 * 6176, np.uint8)
            if rv:  # a retransmission reads the circular buffer the first transmission (rv 0) filled: rm_turbo.c:352-361
                scratch = np.zeros(n_e // 8 + 8, np.uint8)
                assert ref.srsran_rm_turbo_tx_lut(P(w_buff), P(cb_in), P(parity), P(scratch), C.c_uint32(idx), C.c_uint32(n_e), C.c_uint32(0), C.c_uint32(0)) == 0
            assert ref.srsran_rm_turbo_tx_lut(P(w_buff), P(cb_in), P(parity), C.c_void_p(e_bits.ctypes.data + wp // 8), C.c_uint32(idx), C.c_uint32(n_e),
                                              C.c_uint32(wp % 8), C.c_uint32(rv)) == 0
            rp += rlen
            wp += n_e
        key = "tb%d_q%d_rv%d_g%d" % (tbs, Qm, rv, G)
        d[key + "_data"], d[key + "_e"] = data[:tbs // 8], e_bits[:(G + 7) // 8]
        cases.append(key)
        want = np.packbits(O.tb_coded_bits(tbs, Qm, G, rv, None, payload=np.unpackbits(data[:tbs // 8]), tx_order=True)[0])
        assert np.array_equal(want[:wp // 8], e_bits[:wp // 8]), key
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "sch_tx_ref.npz"), **d)
    print("sch_tx_ref.npz", os.path.getsize(os.path.join(OUT, "sch_tx_ref.npz")))


def tcod_lut():
    class Tcod(C.Structure):
        _fields_ = [("max_long_cb", C.c_uint32), ("temp", C.c_void_p)]

    class Crc(C.Structure):
        _fields_ = [("table", C.c_uint64 * 256), ("polynom", C.c_int), ("order", C.c_int), ("crcinit", C.c_uint64), ("crcmask", C.c_uint64),
                    ("crchighbit", C.c_uint64), ("out", C.c_uint32)]

    tc = Tcod()
    assert ref.srsran_tcod_init(C.byref(tc), 6144) == 0
    ref.srsran_rm_turbo_gentables()
    rng = np.random.default_rng(41)
    d, cases = {}, []
    for idx in (0, 9, 60, 120, 187):
        K = ref.srsran_cbsegm_cbsize(idx)
        for with_cb, last in ((0, 1), (1, 0), (1, 1)):
            crc_tb, crc_cb = Crc(), Crc()
            assert ref.srsran_crc_init(C.byref(crc_tb), C.c_uint32(0x1864CFB), 24) == 0 and ref.srsran_crc_init(C.byref(crc_cb), C.c_uint32(0x1800063), 24) == 0
            # a previous block leaves the running transport-block checksum in a non-trivial state
            warm = rng.integers(0, 256, 6144 // 8 + 8).astype(np.uint8)
            wpar = np.zeros(3 * 6144 // 8 + 64, np.uint8)
            ref.srsran_tcod_encode_lut(C.byref(tc), C.byref(crc_tb), C.byref(crc_cb), P(warm), P(wpar), C.c_uint32(187), C.c_bool(False))
            state0 = int(crc_tb.crcinit & crc_tb.crcmask)
            nd = (K - 24 * with_cb - 24 * last) // 8
            if nd < 1:
                continue
            inp = np.zeros(K // 8 + 8, np.uint8)
            inp[:nd] = rng.integers(0, 256, nd)
            before = inp.copy()
            par = np.zeros(K // 4 + 8, np.uint8)
            ret = ref.srsran_tcod_encode_lut(C.byref(tc), C.byref(crc_tb), C.byref(crc_cb) if with_cb else None, P(inp), P(par), C.c_uint32(idx), C.c_bool(bool(last)))
            key = "k%d_c%d_l%d" % (idx, with_cb, last)
            d[key + "_in"], d[key + "_sys"], d[key + "_par"] = before[:nd], inp[:K // 8 + 1], par[:K // 4 + 1]
            d[key + "_meta"] = np.array([idx, K, with_cb, last, state0, int(crc_tb.crcinit & crc_tb.crcmask), ret], dtype=np.int64)
            in_len = 3 * K + 12
            w_buff = np.zeros(3 * 6176, np.uint8)
            outs, pars = [], []
            for rv in (0, 2, 3, 1):
                for out_len, w_off in ((100, 0), (in_len - 5, 3), (in_len + 77, 7), (2 * in_len + 13, 0), (64, 5)):
                    o = np.full((out_len + w_off) // 8 + 3, 0xA5, np.uint8)
                    assert ref.srsran_rm_turbo_tx_lut(P(w_buff), P(inp), P(par), P(o), C.c_uint32(idx), C.c_uint32(out_len), C.c_uint32(w_off), C.c_uint32(rv)) == 0
                    outs.append(o)
                    pars.append([rv, out_len, w_off, o.size])
            d[key + "_tx"], d[key + "_txpar"] = np.concatenate(outs), np.array(pars, dtype=np.int64)
            cases.append(key)
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "tcod_lut_ref.npz"), **d)
    print("tcod_lut_ref.npz", os.path.getsize(os.path.join(OUT, "tcod_lut_ref.npz")))


def modem():
    d = {}
    cases = []
    for mod in range(5):
        for n in (1, 7, 8, 13, 100, 333):
            for scale in (1.0, 40.0):
                x = O.qam_symbols(mod, n, seed=mod * 1000 + n, snr_db=12.0, scale=scale)
                key = "m%d_n%d_s%d" % (mod, n, int(scale))
                d[key + "_x"] = np.array(x)
                for kind, fn in (("s", "srsran_demod_soft_demodulate_s"), ("b", "srsran_demod_soft_demodulate_b"),
                                 ("f", "srsran_demod_soft_demodulate")):
                    out = O.aligned_empty(n * O.QM[mod], O.LLR_DTYPES[kind])
                    assert getattr(ref, fn)(mod, P(x), P(out), n) == 0
                    d[key + "_" + kind] = np.array(out)
                cases.append(key)
    d["cases"] = np.array(cases)
    # scrambling chips: apply to +1 and read the sign
    seqs = []
    for seed, L in ((0, 100), (1, 24), (12345, 1000), (0x7FFFFFFF, 47), ((0x1234 << 14) + (7 << 9) + 301, 40000), (987654321, 150000)):
        one = np.ones(L, np.int16)
        out = np.zeros(L, np.int16)
        ref.srsran_sequence_apply_s(P(one), P(out), C.c_uint32(L), C.c_uint32(seed))
        d["seq_%d_%d" % (seed, L)] = np.packbits(out == -1)
        seqs.append([seed, L])
    d["seqs"] = np.array(seqs, dtype=np.int64)
    ch = []
    for rnti, nslot, cell, q in ((0x1234, 4, 301, 0), (0xFFFF, 19, 503, 1), (1, 0, 0, 0), (70, 13, 150, 1)):
        L = 96
        one = np.ones(L, np.int16)
        o1 = np.zeros(L, np.int16)
        o2 = np.zeros(L, np.int16)
        ref.srsran_sequence_pusch_apply_s(P(one), P(o1), C.c_uint16(rnti), C.c_uint32(nslot), C.c_uint32(cell), C.c_uint32(L))
        ref.srsran_sequence_pdsch_apply_s(P(one), P(o2), C.c_uint16(rnti), C.c_int(q), C.c_uint32(nslot), C.c_uint32(cell), C.c_uint32(L))
        ch.append(np.concatenate([[rnti, nslot, cell, q], np.packbits(o1 == -1), np.packbits(o2 == -1)]))
    d["channel_seeds"] = np.array(ch, dtype=np.int64)
    # srsran_predecoding_single (AVX body + scalar tail), csi = NULL as pusch.c:413 calls it
    ref.srsran_predecoding_single.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_float, C.c_float]
    rng = np.random.default_rng(8)
    n = 333
    y, h = O.aligned_empty(n, np.complex64), O.aligned_empty(n, np.complex64)
    y[:] = rng.normal(size=n) + 1j * rng.normal(size=n)
    h[:] = 0.3 + 0.7 * (rng.normal(size=n) + 1j * rng.normal(size=n))
    xs, pars = [], []
    for scaling, noise in ((1.0, 0.0), (0.7, 0.05), (2.0, 0.3)):
        x = O.aligned_empty(n, np.complex64)
        ref.srsran_predecoding_single(P(y), P(h), P(x), None, n, scaling, noise)
        xs.append(np.array(x))
        pars.append([scaling, noise])
    d["eq_y"], d["eq_h"], d["eq_x"], d["eq_par"] = np.array(y), np.array(h), np.stack(xs), np.array(pars, np.float32)
    np.savez_compressed(os.path.join(OUT, "modem_ref.npz"), **d)
    print("modem_ref.npz", os.path.getsize(os.path.join(OUT, "modem_ref.npz")))


def mod():
    """tests/golden/mod_ref.npz: the reference's modulator -- srsran_modem_table_lte + srsran_modem_table_bytes + srsran_mod_modulate_bytes (mod.c:135-166) on
    byte-packed bits that walk every constellation index, and on seeded bits behind srsran_sequence_apply_packed (sequence.c:609-650)"""
    d = {}
    rng = np.random.default_rng(77)
    for m in range(5):
        qm = O.QM[m]
        t = C.create_string_buffer(4096)
        assert ref.srsran_modem_table_lte(t, m) == 0
        ref.srsran_modem_table_bytes(t)
        idx = np.arange(1 << qm)
        bits = ((idx[:, None] >> (qm - 1 - np.arange(qm))) & 1).astype(np.uint8).reshape(-1)
        if bits.size % 8:
            bits = np.concatenate([bits, np.zeros(8 - bits.size % 8, np.uint8)] * 4)[:8 * qm]  # BPSK / QPSK: a few repeats to fill bytes
        packed = np.packbits(bits)
        n = bits.size // qm
        out = O.aligned_empty(n, np.complex64)
        assert ref.srsran_mod_modulate_bytes(t, P(packed), P(out), C.c_uint32(bits.size)) == n
        d["walk_bits_%d" % m], d["walk_sym_%d" % m] = packed, np.array(out)
        # seeded bits, scrambled in packed form first (pdsch.c:1005-1018)
        nsym = 1000 + 37 * m
        nb = nsym * qm
        raw = np.packbits(rng.integers(0, 2, (nb + 7) // 8 * 8).astype(np.uint8))
        scr = np.zeros_like(raw)
        seed = (0x1234 << 14) + (m << 13) + (3 << 9) + 301
        ref.srsran_sequence_apply_packed(P(raw), P(scr), C.c_uint32(nb), C.c_uint32(seed))
        out = O.aligned_empty(nsym, np.complex64)
        assert ref.srsran_mod_modulate_bytes(t, P(scr), P(out), C.c_uint32(nb)) == nsym
        d["rand_bits_%d" % m], d["rand_sym_%d" % m], d["rand_seed_%d" % m] = raw, np.array(out), np.array([seed, nb], np.int64)
        ref.srsran_modem_table_free(t)
    np.savez_compressed(os.path.join(OUT, "mod_ref.npz"), **d)
    print("mod_ref.npz", os.path.getsize(os.path.join(OUT, "mod_ref.npz")))


def sch_nr():
    """NR transport blocks as sch_nr.c chains the reference's blocks (sch_nr_encode :375-520, sch_nr_decode :522-713; segmentation by
    srsran_cbsegm_ldpc_bg1/2): payload -> e bits -> noisy int8 LLRs -> code-block verdicts, iterations, payload, TB CRC; one case
    with a second transmission (rv 2) for the code blocks the first left undecoded"""
    class Cbsegm(C.Structure):  # srsran_cbsegm_t, cbsegm.h
        _fields_ = [(n, C.c_uint32) for n in ("F", "C", "K1", "K2", "K1_idx", "K2_idx", "C1", "C2", "tbs", "L_tb", "L_cb", "Z")]

    class Args(C.Structure):
        _fields_ = [("type", C.c_int), ("bg", C.c_int), ("ls", C.c_uint16), ("scaling_fctr", C.c_float), ("max_nof_iter", C.c_int)]

    def crc_new(poly, order):
        q = C.create_string_buffer(4096)
        assert ref.srsran_crc_init(q, C.c_uint32(poly), C.c_int(order)) == 0
        return q

    ref.srsran_crc_checksum_byte.restype = C.c_uint32
    crc24a, crc24b, crc16 = crc_new(0x1864CFB, 24), crc_new(0x1800063, 24), crc_new(0x11021, 16)
    QMS = [1, 2, 4, 6, 8]
    rng = np.random.default_rng(2104)
    d, cases = {}, []
    #        tbs    R     mod rv Nl  G      Nref  amp  sigma  max_iter retransmit
    cfgs = ((24, 0.2, 1, 0, 1, 240, 0, 10.0, 8.0, 10, False),
            (3000, 0.5, 1, 0, 1, 6400, 0, 12.0, 9.0, 10, False),
            (3840, 0.5, 2, 1, 1, 8000, 0, 12.0, 7.0, 8, False),
            (20040, 0.8, 3, 0, 2, 26400, 0, 14.0, 5.5, 10, False),
            (20040, 0.8, 3, 0, 1, 25200, 17000, 14.0, 4.0, 6, False),
            (50184, 0.75, 4, 0, 1, 67200, 0, 16.0, 5.0, 10, False),
            (50184, 0.9, 4, 0, 1, 55680, 0, 16.0, 6.8, 4, True))
    for tbs, R, mod, rv, Nl, G, Nref_in, amp, sigma, max_iter, retx in cfgs:
        bg = 1 if (tbs <= 292 or (tbs <= 3824 and R <= 0.67) or R <= 0.25) else 0
        seg = Cbsegm()
        assert (ref.srsran_cbsegm_ldpc_bg2 if bg else ref.srsran_cbsegm_ldpc_bg1)(C.byref(seg), C.c_uint32(tbs)) == 0
        Cn, Z, Kr, L_tb, L_cb, Qm = seg.C, seg.Z, seg.K1, seg.L_tb, seg.L_cb, QMS[mod]
        Bp = tbs + L_tb + L_cb * Cn
        assert Bp % Cn == 0
        Kp = Bp // Cn
        F = Kr - Kp
        N = Z * (66 if bg == 0 else 50)
        Nref = Nref_in if Nref_in else N
        crc_tb = crc24a if L_tb == 24 else crc16

        def get_E(j):
            q = Nl * Qm
            return q * (G // (q * Cn)) if j <= Cn - (G // q) % Cn - 1 else q * -(-G // (q * Cn))

        payload = rng.integers(0, 256, tbs // 8).astype(np.uint8)
        checksum_tb = ref.srsran_crc_checksum_byte(crc_tb, P(payload), C.c_int(tbs))
        enc = C.create_string_buffer(256)
        assert ref.srsran_ldpc_encoder_init(enc, 0, bg, C.c_uint16(Z)) == 0
        rmt = C.create_string_buffer(64)
        assert ref.srsran_ldpc_rm_tx_init(rmt) == 0
        bits = np.unpackbits(payload)

        def transmit(rv_):
            out, inp = [], 0
            for r in range(Cn):
                cb_len = Kp - L_cb - (L_tb if r == Cn - 1 else 0)
                cb = np.zeros(Kr, np.uint8)
                cb[:cb_len] = bits[inp:inp + cb_len]
                if r == Cn - 1:
                    cb[cb_len:cb_len + L_tb] = [(checksum_tb >> (L_tb - 1 - i)) & 1 for i in range(L_tb)]
                inp += cb_len // 8 * 8
                if L_cb:
                    ref.srsran_crc_attach(crc24b, P(cb), C.c_int(Kp - L_cb))
                cb[Kp:] = 254
                cw = np.zeros(N, np.uint8)
                assert ref.srsran_ldpc_encoder_encode(enc, P(cb), P(cw), C.c_uint32(Kr)) == 0
                E = get_E(r)
                tx = np.zeros(E, np.uint8)
                ref.srsran_ldpc_rm_tx(rmt, P(cw), P(tx), C.c_uint32(E), C.c_int(bg), C.c_uint32(Z), C.c_uint8(rv_), C.c_int(mod), C.c_uint32(Nref))
                out.append(tx)
            return out

        def channel(tx):
            y = amp * (1.0 - 2.0 * tx) + sigma * rng.standard_normal(tx.size)
            return np.clip(np.round(y), -63, 63).astype(np.int8)

        dec = C.create_string_buffer(4096)
        a = Args(2, bg, Z, 0.8, max_iter)
        assert ref.srsran_ldpc_decoder_init(dec, C.byref(a)) == 0
        rmr = C.create_string_buffer(64)
        assert ref.srsran_ldpc_rm_rx_init_c(rmr) == 0
        softbuf = np.zeros((Cn, N), np.int8)
        cb_crc = np.zeros(Cn, np.uint8)
        cb_data = np.zeros((Cn, (Kr + 7) // 8), np.uint8)
        key = "tbs%d_m%d_rv%d_g%d_n%d" % (tbs, mod, rv, G, Nref_in)
        rounds = [rv] + ([2] if retx else [])
        for ti, rv_ in enumerate(rounds):
            segs = transmit(rv_)
            llr_all = np.concatenate([channel(segs[r]) for r in range(Cn) if not cb_crc[r]] + [np.zeros(0, np.int8)])
            crc_before = cb_crc.copy()
            inp, it_sum, its = 0, 0, []
            for r in range(Cn):
                E = get_E(r)
                if cb_crc[r]:
                    its.append(-1)
                    continue
                n_llr = ref.srsran_ldpc_rm_rx_c(rmr, P(llr_all[inp:inp + E].copy()), P(softbuf[r]), C.c_uint32(E), C.c_uint32(F), C.c_int(bg), C.c_uint32(Z),
                                                C.c_uint8(rv_), C.c_int(mod), C.c_uint32(Nref))
                assert n_llr > 0
                crc = crc24b if L_cb else crc_tb
                temp = np.zeros(Kr, np.uint8)
                ret = ref.srsran_ldpc_decoder_decode_crc_c(dec, P(softbuf[r]), P(temp), C.c_uint32(n_llr), crc)
                assert ret >= 0
                its.append(ret)
                it_sum += max_iter if ret == 0 else ret
                cb_len = Kp - L_cb
                cb_crc[r] = 1 if (ret != 0 and temp[:cb_len].any()) else 0
                if cb_crc[r]:
                    cb_data[r, :cb_len // 8] = np.packbits(temp[:cb_len])
                inp += E
            out = np.zeros(tbs // 8, np.uint8)
            crc_ok = 0
            if cb_crc.all():
                parts = []
                for r in range(Cn):
                    cb_len = Kp - L_cb - (L_tb if r == Cn - 1 else 0)
                    parts.append(cb_data[r, :cb_len // 8])
                out = np.concatenate(parts)
                if Cn == 1:
                    crc_ok = 1
                else:
                    last = np.unpackbits(cb_data[Cn - 1])[Kp - L_cb - L_tb:Kp - L_cb]
                    c2 = int("".join(map(str, last)), 2)
                    crc_ok = int(ref.srsran_crc_checksum_byte(crc_tb, P(out), C.c_int(tbs)) == c2)
            k = key + "_t%d" % ti
            d[k + "_llr"], d[k + "_crc_in"], d[k + "_crc_out"], d[k + "_its"] = llr_all, crc_before, cb_crc.copy(), np.array(its, np.int32)
            d[k + "_res"] = np.array([crc_ok, it_sum], np.int32)
            d[k + "_out"] = out
            d[k + "_e"] = np.packbits(np.concatenate(segs))
            d[k + "_soft_crc"] = np.array([zlib.crc32(softbuf.tobytes())], np.int64)
        d[key + "_par"] = np.array([tbs, int(round(R * 1000)), mod, rv, Nl, G, Nref, max_iter, Cn, Z, Kr, Kp, F, L_tb, L_cb, bg, len(rounds)], np.int32)
        d[key + "_payload"] = payload
        cases.append(key)
        ref.srsran_ldpc_decoder_free(dec)
        print(key, "C", Cn, "Z", Z, "F", F, "crc", [int(x) for x in d[key + "_t%d_res" % (len(rounds) - 1)]], "cb", d[key + "_t%d_crc_out" % (len(rounds) - 1)].tolist(),
              [d[key + "_t%d_its" % t].tolist() for t in range(len(rounds))])
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "sch_nr_ref.npz"), **d)
    print("sch_nr_ref.npz", os.path.getsize(os.path.join(OUT, "sch_nr_ref.npz")))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["turbo", "turbo8", "ldpc", "ldpc_fs", "syncglue", "rm", "modem", "ldpc_tx", "sch_tx", "ldpc_flood", "tcod_lut", "sch_nr", "sync_captures", "ref_link", "mod"]
    for name in which:
        {"turbo": turbo, "turbo8": turbo8, "ldpc": ldpc, "ldpc_fs": ldpc_fs, "syncglue": syncglue, "rm": rm, "modem": modem, "ldpc_tx": ldpc_tx, "sch_tx": sch_tx, "ldpc_flood": ldpc_flood, "tcod_lut": tcod_lut, "sch_nr": sch_nr, "sync_captures": sync_captures, "ref_link": ref_link, "mod": mod}[name]()
