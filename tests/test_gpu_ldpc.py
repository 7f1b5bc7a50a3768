"""Parity of the HIP LDPC decoder with the oracle (restated ldpc_dec_c.c), through the C ABI. Bar: bit-exact."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu

ALL_LS = [z for z in range(2, 385) if O.orc().orc_ldpc_ls_index(z) >= 0]


@pytest.mark.parametrize("bg", [0, 1])
def test_all_lifting_sizes(hiplib, bg):
    """all 51 lifting sizes (the reference's own test grid, ldpc/test/CMakeLists.txt:69-145)"""
    import srslte_amd as S

    assert len(ALL_LS) == 51
    for i, Z in enumerate(ALL_LS):
        snr, sf, nit = ((2.0, 0.8, 10), (0.0, 0.75, 5), (-2.0, 0.8, 3))[i % 3]
        n_cw = 5 if Z > 64 else 13
        msgs, llrs = O.ldpc_llrs(bg, Z, n_cw, snr, seed=Z * 2 + bg, clip=127 if i % 2 else 63)
        ref, _ = O.ldpc_decode(bg, Z, llrs, sf, nit)
        out = S.LdpcBatch(bg, Z, sf, nit, n_cw).decode(llrs)
        assert np.array_equal(ref, out), "BG%d Z=%d: %d words differ" % (bg + 1, Z, np.any(ref != out, axis=1).sum())


@pytest.mark.parametrize("sf", [0.8, 0.75])
@pytest.mark.parametrize("regime", ["stuck", "late"])
@pytest.mark.parametrize("bg", [0, 1])
def test_z384_at_20_iterations_noisy(hiplib, bg, regime, sf):
    """BASELINE configs[2]'s regime: Z = 384 at the full 20 iterations on NOISY words -- "stuck": just below the decoding threshold
    of the mother code (BG1 -1.5 dB, BG2 -3.8 dB: words never converge, every iteration changes messages); "late": just above it
    (-1.0 / -3.5 dB: convergence between iteration 10 and 20) -- with both scaling factors the reference uses (sch_nr.c:275 0.8,
    ldpc_chain_test.c:76 0.75): final messages from both kernels, and the messages after iterations 1, 7, 13, 20, equal the oracle's"""
    import srslte_amd as S

    n_cw = 6
    snr = {(0, "stuck"): -1.5, (0, "late"): -1.0, (1, "stuck"): -3.8, (1, "late"): -3.5}[(bg, regime)]
    msgs, llrs = O.ldpc_llrs(bg, 384, n_cw, snr, seed=1)
    ref, rets = O.ldpc_decode(bg, 384, llrs, sf, 20)
    assert rets == [20] * n_cw
    dec = S.LdpcBatch(bg, 384, sf, 20, n_cw)
    out = dec.decode(llrs)  # the two-positions-per-lane kernel (what bench.py's extra.ldpc times)
    assert np.array_equal(ref, out), "BG%d: %d words differ" % (bg + 1, np.any(ref != out, axis=1).sum())
    out, per_it = dec.decode(llrs, want_iter_msgs=True)  # the one-position-per-lane kernel, with snapshots
    assert np.array_equal(ref, out)
    K = dec.liftK
    snaps = {}
    for k in (1, 7, 13, 20):
        snaps[k], _ = O.ldpc_decode(bg, 384, llrs, sf, k)
        assert np.array_equal(np.unpackbits(per_it[:, k - 1], axis=1)[:, :K], snaps[k]), (bg, snr, sf, k)
    good = (ref == msgs).all(axis=1).sum()
    if sf == 0.8:  # the regimes were located with this scaling
        if regime == "stuck":
            assert good < n_cw  # not everything decodes even at 20 iterations
        else:
            assert good == n_cw and (snaps[7] == msgs).all(axis=1).sum() < n_cw  # the later iterations did the work
    assert not np.array_equal(snaps[13], snaps[7])  # the iterations are still moving bits


@pytest.mark.parametrize("bg,Z", [(0, 384), (1, 384), (0, 96), (1, 30)])
def test_rate_matched_lengths_and_scalings(hiplib, bg, Z):
    """cdwd_rm_length clamps of ldpc_decoder.c:51-65 and several scaling factors"""
    import srslte_amd as S

    g = O.ldpc_graph(bg, Z)
    N = g.bgN * Z
    msgs, llrs = O.ldpc_llrs(bg, Z, 4, 1.0, seed=Z)
    for rm in (N - 2 * Z, N, 5, (g.bgK + 2) * Z, (g.bgK + 7) * Z + 1, (g.bgK + 20) * Z - 1):
        for sf in (0.8, 0.75, 0.5, 1.0, 0.3):
            ref, _ = O.ldpc_decode(bg, Z, llrs, sf, 7, rm)
            out = S.LdpcBatch(bg, Z, sf, 7, 4).decode(llrs, rm)
            assert np.array_equal(ref, out), (bg, Z, rm, sf)


def test_default_iterations_and_errors(hiplib):
    import srslte_amd as S
    from srslte_amd import capi

    msgs, llrs = O.ldpc_llrs(0, 64, 3, 0.5, seed=1)
    ref, rets = O.ldpc_decode(0, 64, llrs, 0.8, 0)  # 0 -> default 10 (ldpc_decoder.c:42,579)
    assert rets == [10] * 3
    assert np.array_equal(ref, S.LdpcBatch(0, 64, 0.8, 0, 3).decode(llrs))
    h = C.c_void_p()
    lib = S.lib()
    assert lib.srsran_hip_ldpc_batch_create(C.byref(h), 0, 17, 0.8, 10, 1) == capi.SRSRAN_ERROR_INVALID_INPUTS
    assert lib.srsran_hip_ldpc_batch_create(C.byref(h), 0, 16, 1.5, 10, 1) == capi.SRSRAN_ERROR_INVALID_INPUTS
    assert lib.srsran_hip_ldpc_batch_create(C.byref(h), 2, 16, 0.8, 10, 1) == capi.SRSRAN_ERROR_INVALID_INPUTS


def test_handle_api_and_crc_early_stop(hiplib):
    """srsran_ldpc_decoder_init / decode_c / decode_crc_c (CRC24B early stop, ldpc_decoder.c:87-99)"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    poly, order = 0x1800063, 24  # CRC24B as used for NR code blocks
    for bg, Z in ((0, 384), (1, 128), (0, 36)):
        g = O.ldpc_graph(bg, Z)
        K, N = g.bgK * Z, g.bgN * Z
        rng = np.random.default_rng(Z)
        q = capi.LdpcDecoder()
        args = capi.LdpcDecoderArgs(capi.LDPC_C_AVX2, bg, Z, 0.8, 12)
        assert lib.srsran_ldpc_decoder_init(C.byref(q), C.byref(args)) == 0
        assert (q.bgN, q.bgM, q.bgK, q.liftK, q.liftN) == (g.bgN, g.bgM, g.bgK, K, N)
        crc = capi.Crc()
        crc.polynom, crc.order = poly, order
        for snr in (3.0, 0.5, -3.0):
            msg = rng.integers(0, 2, K).astype(np.uint8)
            c = O.orc().orc_crc_bits(poly, order, O.P(msg), K - order)
            msg[K - order:] = [(c >> (order - 1 - i)) & 1 for i in range(order)]
            cw = np.zeros(N - 2 * Z, np.uint8)
            assert O.orc().orc_ldpc_encode(C.byref(g), O.P(msg), O.P(cw)) == 0
            sigma = 10 ** (-snr / 20)
            llr = np.clip(np.round(((1.0 - 2.0 * cw) + sigma * rng.standard_normal(cw.size)) * 8 / sigma ** 2), -63, 63).astype(np.int8)
            ref, rets = O.ldpc_decode(bg, Z, llr[None], 0.8, 12, None, crc=(poly, order))
            out = np.zeros(K, np.uint8)
            ret = lib.srsran_ldpc_decoder_decode_crc_c(C.byref(q), O.P(llr), O.P(out), N - 2 * Z, C.byref(crc))
            assert ret == rets[0], (bg, Z, snr, ret, rets)
            if ret > 0:
                assert np.array_equal(out, ref[0]) and np.array_equal(out, msg)
            ref2, _ = O.ldpc_decode(bg, Z, llr[None], 0.8, 12)
            out2 = np.zeros(K, np.uint8)
            assert lib.srsran_ldpc_decoder_decode_c(C.byref(q), O.P(llr), O.P(out2), N - 2 * Z) == 12
            assert np.array_equal(out2, ref2[0])
        lib.srsran_ldpc_decoder_free(C.byref(q))
        assert not q.ptr
    # a decoder type that does not exist (ldpc_decoder.c:644-646)
    q = capi.LdpcDecoder()
    args = capi.LdpcDecoderArgs(8, 0, 16, 0.8, 10)
    assert lib.srsran_ldpc_decoder_init(C.byref(q), C.byref(args)) == -1


def test_compact_pcm_export(hiplib):
    """create_compact_pcm (base_graph.c:4467-4503) against the oracle's graph"""
    import srslte_amd as S

    for bg, (M, N) in enumerate(((46, 68), (42, 52))):
        for Z in (2, 7, 120, 384):
            pcm = np.zeros(M * N, np.uint16)
            pos = np.zeros((M, 20), np.int8)
            assert S.lib().create_compact_pcm(O.P(pcm), O.P(pos), bg, Z) == 0
            g = O.ldpc_graph(bg, Z)
            pcm = pcm.reshape(M, N)
            assert (pcm != 0xFFFF).sum() == g.nof_edges
            for m in range(M):
                cols = [g.col[e] for e in range(g.row_start[m], g.row_start[m + 1])]
                assert [int(c) for c in pos[m] if c >= 0] == cols
                assert [int(pcm[m, c]) for c in cols] == [g.shift[e] for e in range(g.row_start[m], g.row_start[m + 1])]


def test_full_size_roundtrip_property(hiplib):
    """BASELINE size: thousands of BG1 Z=384 words: noise-free round trip, identical copies decode identically"""
    import srslte_amd as S

    n_cw = 2048
    msgs, llrs = O.ldpc_llrs(0, 384, 4, 30.0, seed=2)
    big = np.tile(llrs, (n_cw // 4, 1))
    out = S.LdpcBatch(0, 384, 0.8, 20, n_cw).decode(big)
    assert np.array_equal(out, np.tile(msgs, (n_cw // 4, 1)))


def _fs_llrs(bg, Z, n_cw, snr, seed, kind):
    """float32 / int16 LLRs from the same channel as the int8 ones; int16 ones large enough to reach the 15-bit clip"""
    _, l8 = O.ldpc_llrs(bg, Z, n_cw, snr, seed=seed, clip=127)
    rng = np.random.default_rng(seed)
    if kind == "f":
        return (l8 * rng.uniform(0.3, 0.35, l8.shape)).astype(np.float32)
    return (l8.astype(np.int32) * (180 if seed % 2 else 3)).clip(-32767, 32767).astype(np.int16)


@pytest.mark.parametrize("kind", ["f", "s"])
@pytest.mark.parametrize("bg", [0, 1])
def test_float_and_int16_decoders_all_lifting_sizes(hiplib, bg, kind):
    """SRSRAN_LDPC_DECODER_F / _S (ldpc_dec_f.c, ldpc_dec_s.c): bit-exact messages AND bit-exact a-posteriori soft
    bits (the float decoder only uses IEEE subtract / multiply / add, evaluated unfused on the device)"""
    import srslte_amd as S
    from srslte_amd import capi

    typ = capi.LDPC_F if kind == "f" else capi.LDPC_S
    for i, Z in enumerate(ALL_LS):
        snr, sf, nit = ((2.0, 0.8, 10), (0.0, 0.75, 5), (-2.0, 0.8, 3))[i % 3]
        n_cw = 3 if Z > 64 else 7
        llrs = _fs_llrs(bg, Z, n_cw, snr, Z * 2 + bg, kind)
        ref, ref_soft = O.ldpc_decode_fs(bg, Z, llrs, sf, nit, want_soft=True)
        out, soft = S.LdpcBatch(bg, Z, sf, nit, n_cw, typ).decode(llrs, want_soft=True)
        assert np.array_equal(ref, out), "BG%d Z=%d: %d words differ" % (bg + 1, Z, np.any(ref != out, axis=1).sum())
        assert np.array_equal(ref_soft.view(np.uint8), soft.view(np.uint8)), "BG%d Z=%d soft bits" % (bg + 1, Z)


@pytest.mark.parametrize("kind", ["f", "s"])
def test_float_and_int16_rate_matched_and_handle(hiplib, kind):
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    typ = capi.LDPC_F if kind == "f" else capi.LDPC_S
    bg, Z = 0, 384
    g = O.ldpc_graph(bg, Z)
    N = g.bgN * Z
    llrs = _fs_llrs(bg, Z, 3, 1.0, 7, kind)
    for rm in (N - 2 * Z, 5, (g.bgK + 7) * Z + 1):
        for sf in (0.8, 0.5, 1.0):
            ref = O.ldpc_decode_fs(bg, Z, llrs, sf, 6, rm)
            assert np.array_equal(ref, S.LdpcBatch(bg, Z, sf, 6, 3, typ).decode(llrs, rm)), (rm, sf)
    # drop-in handle: one decode entry point per object type (ldpc_decoder.c:170-260)
    q = capi.LdpcDecoder()
    args = capi.LdpcDecoderArgs(typ, bg, Z, 0.8, 6)
    assert lib.srsran_ldpc_decoder_init(C.byref(q), C.byref(args)) == 0
    dec = lib.srsran_ldpc_decoder_decode_f if kind == "f" else lib.srsran_ldpc_decoder_decode_s
    other = lib.srsran_ldpc_decoder_decode_s if kind == "f" else lib.srsran_ldpc_decoder_decode_f
    ref = O.ldpc_decode_fs(bg, Z, llrs, 0.8, 6)
    for i in range(3):
        msg = np.zeros(g.bgK * Z, np.uint8)
        assert dec(C.byref(q), O.P(llrs[i]), O.P(msg), N - 2 * Z) == 6
        assert np.array_equal(msg, ref[i])
    assert other(C.byref(q), O.P(llrs[0]), O.P(msg), N - 2 * Z) == -1
    assert lib.srsran_ldpc_decoder_decode_c(C.byref(q), O.P(llrs[0]), O.P(msg), N - 2 * Z) == -1
    lib.srsran_ldpc_decoder_free(C.byref(q))
    # wrong entry point for the batch object type
    b = S.LdpcBatch(bg, Z, 0.8, 6, 1, typ)
    d = S.DeviceBuffer(N * 4)
    assert lib.srsran_hip_ldpc_batch_run(b._h, d.ptr, N, d.ptr, N, 1, N, None, None) == capi.SRSRAN_ERROR_INVALID_INPUTS
    h = C.c_void_p()
    assert lib.srsran_hip_ldpc_batch_create_typed(C.byref(h), 0, 16, 0.8, 10, 1, 8) == capi.SRSRAN_ERROR_INVALID_INPUTS


@pytest.mark.parametrize("bg,Z", [(0, 384), (1, 208), (0, 36), (1, 7), (0, 2), (1, 128), (0, 104)])
def test_flooded_schedule(hiplib, bg, Z):
    """SRSRAN_LDPC_DECODER_C_FLOOD (ldpc_dec_c_flood.c, ldpc_decoder.c:105-160): messages AND a-posteriori soft bits bit-exact,
    full and shortened code words, CRC early stop through the handle"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    g = O.ldpc_graph(bg, Z)
    K, N = g.bgK * Z, g.bgN * Z
    n_cw = 5
    for snr, clip, nit in ((2.0, 63, 4), (0.5, 127, 7)):
        _, llrs = O.ldpc_llrs(bg, Z, n_cw, snr, seed=Z * 3 + bg, clip=clip)
        for typ in (capi.LDPC_C_FLOOD, capi.LDPC_C_AVX2_FLOOD):
            dec = S.LdpcBatch(bg, Z, 0.8, nit, n_cw, typ)
            for rm in (N - 2 * Z, (g.bgK + 9) * Z + 3, (g.bgK + 2) * Z):
                out, soft = dec.decode(llrs, cdwd_rm_length=rm, want_soft=True)
                for i in range(n_cw):
                    want, ws, ret = O.ldpc_decode_flood(bg, Z, llrs[i], 0.8, nit, rm)
                    assert ret == nit
                    assert np.array_equal(out[i], want), (typ, rm, i)
                    assert np.array_equal(soft[i], ws), (typ, rm, i)
    # handle: decode_c and decode_crc_c (iteration count of the first CRC match, out of 2 * max_nof_iter)
    q = capi.LdpcDecoder()
    args = capi.LdpcDecoderArgs(capi.LDPC_C_FLOOD, bg, Z, 0.8, 6)
    assert lib.srsran_ldpc_decoder_init(C.byref(q), C.byref(args)) == 0
    msg = np.zeros(K, np.uint8)
    assert lib.srsran_ldpc_decoder_decode_c(C.byref(q), O.P(llrs[0]), O.P(msg), N - 2 * Z) == 6
    assert np.array_equal(msg, O.ldpc_decode_flood(bg, Z, llrs[0], 0.8, 6, N - 2 * Z)[0])
    if K >= 40:
        poly, order = 0x1800063, 24
        crc = capi.Crc()
        crc.polynom, crc.order = poly, order
        rng = np.random.default_rng(Z)
        for snr in (3.0, 0.0):
            m = rng.integers(0, 2, K).astype(np.uint8)
            cs = O.orc().orc_crc_bits(poly, order, O.P(m), K - order)
            m[K - order:] = [(cs >> (order - 1 - i)) & 1 for i in range(order)]
            cw = np.zeros(N - 2 * Z, np.uint8)
            assert O.orc().orc_ldpc_encode(C.byref(g), O.P(m), O.P(cw)) == 0
            sigma = 10 ** (-snr / 20)
            llr = np.clip(np.round(((1.0 - 2.0 * cw) + sigma * rng.standard_normal(cw.size)) * 8 / sigma ** 2), -63, 63).astype(np.int8)
            want, _, ret = O.ldpc_decode_flood(bg, Z, llr, 0.8, 6, N - 2 * Z, crc=(poly, order))
            got = lib.srsran_ldpc_decoder_decode_crc_c(C.byref(q), O.P(llr), O.P(msg), N - 2 * Z, C.byref(crc))
            assert got == ret, (bg, Z, snr, got, ret)
            if ret:
                assert np.array_equal(msg, want)
    lib.srsran_ldpc_decoder_free(C.byref(q))


@pytest.mark.parametrize("bg,Z,typ", [(0, 384, "C"), (1, 208, "C"), (0, 36, "C"), (1, 16, "C"), (0, 7, "C"), (0, 96, "FLOOD"), (1, 30, "FLOOD")])
def test_batch_crc_early_stop(hiplib, bg, Z, typ):
    """srsran_hip_ldpc_batch_run_crc: per code word the iteration count of the first CRC match (0 = none) and the message the
    reference returns, code words of very different SNR in one launch (several per workgroup for small Z)"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    g = O.ldpc_graph(bg, Z)
    K, N = g.bgK * Z, g.bgN * Z
    rng = np.random.default_rng(Z + bg)
    for poly, order in ((0x1800063, 24), (0x11021, 16)):
        if K <= order + 8:
            continue
        n_cw, nit = 24, 8
        llrs = np.zeros((n_cw, N - 2 * Z), np.int8)
        for i in range(n_cw):
            m = rng.integers(0, 2, K).astype(np.uint8)
            cs = O.orc().orc_crc_bits(poly, order, O.P(m), K - order)
            m[K - order:] = [(cs >> (order - 1 - j)) & 1 for j in range(order)]
            cw = np.zeros(N - 2 * Z, np.uint8)
            assert O.orc().orc_ldpc_encode(C.byref(g), O.P(m), O.P(cw)) == 0
            snr = (4.0, 1.5, 0.0, -4.0)[i % 4]
            sigma = 10 ** (-snr / 20)
            llrs[i] = np.clip(np.round(((1.0 - 2.0 * cw) + sigma * rng.standard_normal(cw.size)) * 8 / sigma ** 2), -63, 63).astype(np.int8)
        flood = typ == "FLOOD"
        dec = S.LdpcBatch(bg, Z, 0.8, nit, n_cw, capi.LDPC_C_FLOOD if flood else capi.LDPC_C_AVX2)
        d_llr, d_msg = S.DeviceBuffer.from_numpy(llrs), S.DeviceBuffer(n_cw * K)
        d_it = S.DeviceBuffer.from_numpy(np.full(n_cw, -1, np.int32))
        for rm in (N - 2 * Z, (g.bgK + 12) * Z):
            capi.check(lib.srsran_hip_ldpc_batch_run_crc(dec._h, d_llr.ptr, N - 2 * Z, d_msg.ptr, K, n_cw, rm, poly, order, d_it.ptr, None), "run_crc")
            capi.check(lib.srsran_hip_stream_sync(None), "sync")
            its, msg = d_it.to_numpy(np.int32, (n_cw,)), d_msg.to_numpy(np.uint8, (n_cw, K))
            seen = set()
            for i in range(n_cw):
                if flood:
                    want, _, ret = O.ldpc_decode_flood(bg, Z, llrs[i], 0.8, nit, rm, crc=(poly, order))
                else:
                    w, r = O.ldpc_decode(bg, Z, llrs[i][None], 0.8, nit, rm, crc=(poly, order))
                    want, ret = w[0], r[0]
                assert its[i] == ret, (typ, rm, i, its[i], ret)
                if ret:
                    assert np.array_equal(msg[i], want), (typ, rm, i)
                seen.add(ret)
            assert len(seen) >= 2  # the batch really mixes early and late / failed code words
    bad = S.LdpcBatch(bg, Z, 0.8, 4, 1, capi.LDPC_S)
    assert lib.srsran_hip_ldpc_batch_run_crc(bad._h, d_llr.ptr, N, d_msg.ptr, K, 1, N, 0x1800063, 24, d_it.ptr, None) == capi.SRSRAN_ERROR_INVALID_INPUTS

@pytest.mark.parametrize("bg,Z", [(0, 384), (1, 208), (0, 36), (1, 16), (0, 320)])
def test_extreme_and_full_range_inputs(hiplib, bg, Z):
    """channel LLRs made of nothing but the values around the decoder's special cases (-128, +-127 = infinity, +-64, +-63, 0), uniform noise over
    the whole int8 range, and weak LLRs salted with -128: the packed kernel keeps infinity as +-64 and its bytes biased by 128 inside, none of
    which may show (ldpc_dec_c.c:170-188, 308-315, 338-363)"""
    import srslte_amd as S

    rng = np.random.default_rng(5 + Z)
    g = O.ldpc_graph(bg, Z)
    n = g.bgN * Z - 2 * Z
    llrs = np.stack([rng.choice(np.array([-128, -127, -64, -63, 63, 64, 127, 0], np.int8), n) for _ in range(4)] +
                    [rng.integers(-128, 128, n).astype(np.int8) for _ in range(4)] +
                    [np.where(rng.random(n) < 0.1, -128, rng.integers(-30, 31, n)).astype(np.int8) for _ in range(4)])
    for nit in (1, 3, 10):
        want, _ = O.ldpc_decode(bg, Z, llrs, 0.8, nit)
        got = S.LdpcBatch(bg, Z, 0.8, nit, llrs.shape[0]).decode(llrs)
        assert np.array_equal(np.asarray(got), np.asarray(want)), (bg, Z, nit)
