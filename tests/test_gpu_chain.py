"""Whole bit-level receive chains on the device against the same chain of oracle functions:
LTE  equaliser -> soft demodulation + descrambling -> rate de-matching -> turbo decoding with CRC early stop -> TB CRC
NR   soft demodulation + sign change + descrambling -> LDPC rate de-matching -> LDPC decoding with CRC early stop
(the transmit sides that make the test signals run on the device too)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def test_lte_pusch_chain(hiplib):
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(17)
    cases = [(6200, 1, 2, 7800), (12960, 2, 4, 18000), (31704, 3, 6, 42000), (75376, 3, 6, 100800)]  # tbs, mod, Qm, G
    n_tb = len(cases)
    enc, dem, sch = C.c_void_p(), C.c_void_p(), C.c_void_p()
    capi.check(lib.srsran_hip_sch_enc_create(C.byref(enc)), "enc")
    capi.check(lib.srsran_hip_demod_create(C.byref(dem)), "dem")
    capi.check(lib.srsran_hip_sch_create(C.byref(sch)), "sch")
    payload = [rng.integers(0, 256, c[0] // 8).astype(np.uint8) for c in cases]
    off_b = np.concatenate([[0], np.cumsum([p.size for p in payload])]).astype(int)
    off_e = np.concatenate([[0], np.cumsum([c[3] for c in cases])]).astype(int)
    ncb = [O.cbsegm(c[0])["C"] for c in cases]
    first = np.concatenate([[0], np.cumsum(ncb)]).astype(int)
    d_pay = S.DeviceBuffer.from_numpy(np.concatenate(payload))
    d_tx = S.DeviceBuffer(int(off_e[-1]) // 8 + 8)
    tx = (capi.HipTb * n_tb)(*[capi.HipTb(c[0], c[2], 0, c[3], int(off_e[i]), int(off_b[i]), 0) for i, c in enumerate(cases)])
    capi.check(lib.srsran_hip_sch_encode(enc, d_pay.ptr, tx, n_tb, d_tx.ptr, None), "encode")
    capi.check(lib.srsran_hip_stream_sync(None), "sync")
    e = np.unpackbits(d_tx.to_numpy(np.uint8, (int(off_e[-1]) // 8 + 8,)))
    # channel: scramble, map, flat fading + noise per transport block (host side: this is the test signal, not the product)
    y_all, h_all, jobs, sym_off, seeds = [], [], [], 0, []
    for i, (tbs, mod, Qm, G) in enumerate(cases):
        seed = O.pusch_seed(0x30 + i, 2 * i, 77)
        bits = e[off_e[i]:off_e[i + 1]] ^ O.sequence_bits(seed, G)
        x = O.modulate(bits, mod)
        h = (0.8 + 0.3 * rng.standard_normal(x.size) + 0.3j * rng.standard_normal(x.size)).astype(np.complex64)
        sigma = 10 ** (-(16.0 + 5.5 * (mod - 1)) / 20) / np.sqrt(2)
        y = (h * x + sigma * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
        y_all.append(y)
        h_all.append(h)
        jobs.append(capi.HipDemodJob(mod, x.size, sym_off, int(off_e[i]), seed, 1))
        seeds.append(seed)
        sym_off += x.size
    y_all, h_all = np.concatenate(y_all), np.concatenate(h_all)
    d_y, d_h, d_x = S.DeviceBuffer.from_numpy(y_all), S.DeviceBuffer.from_numpy(h_all), S.DeviceBuffer(y_all.size * 8)
    d_llr = S.DeviceBuffer(int(off_e[-1]) * 2)
    noise = 0.02
    capi.check(lib.srsran_hip_predecoding_single(d_y.ptr, d_h.ptr, d_x.ptr, None, y_all.size, 1.0, noise, None), "eq")
    capi.check(lib.srsran_hip_demod_run(dem, d_x.ptr, d_llr.ptr, capi.LLR_SHORT, (capi.HipDemodJob * n_tb)(*jobs), n_tb, None), "demod")
    dlen = [c[0] // 8 + 8 for c in cases]
    off_d = np.concatenate([[0], np.cumsum(dlen)]).astype(int)
    d_out = S.DeviceBuffer.from_numpy(np.zeros(int(off_d[-1]), np.uint8))
    d_soft = S.DeviceBuffer.from_numpy(np.zeros((int(first[-1]), capi.SOFTBUFFER_CB_SIZE), np.int16))
    flags = np.zeros(int(first[-1]), np.uint8)
    res = (capi.HipTbResult * n_tb)()
    rx = (capi.HipTb * n_tb)(*[capi.HipTb(c[0], c[2], 0, c[3], int(off_e[i]), int(off_d[i]), int(first[i])) for i, c in enumerate(cases)])
    capi.check(lib.srsran_hip_sch_decode(sch, d_llr.ptr, rx, n_tb, 8, d_soft.ptr, flags.ctypes.data, d_out.ptr, res, None), "decode")
    out = d_out.to_numpy(np.uint8, (int(off_d[-1]),))
    x_dev = d_x.to_numpy(np.complex64, (y_all.size,))
    llr_dev = d_llr.to_numpy(np.int16, (int(off_e[-1]),))
    # the oracle chain on the equaliser output of the device (float stage: 1e-6) and on its own
    x_orc = O.predecoding_single(y_all, h_all, 1.0, noise)
    assert np.abs(x_dev - x_orc).max() <= 1e-6 * np.abs(x_orc).max()
    so = 0
    for i, (tbs, mod, Qm, G) in enumerate(cases):
        n = G // Qm
        llr = O.sequence_apply(O.demod_soft(mod, x_dev[so:so + n], "s"), seeds[i])
        assert np.array_equal(llr_dev[off_e[i]:off_e[i + 1]], llr), i
        soft, crc = np.zeros((ncb[i], capi.SOFTBUFFER_CB_SIZE), np.int16), np.zeros(ncb[i], np.uint8)
        ret, data, avg = O.sch_decode_tb(tbs, Qm, 0, llr, soft, crc, 8)
        assert ret == res[i].crc_ok and abs(avg - res[i].avg_iterations) < 1e-6, i
        assert ret == 0, "test signal too noisy for case %d" % i
        assert np.array_equal(out[off_d[i]:off_d[i] + tbs // 8], payload[i]) and np.array_equal(data[:tbs // 8], payload[i])
        so += n
    lib.srsran_hip_sch_enc_free(enc)
    lib.srsran_hip_demod_free(dem)
    lib.srsran_hip_sch_free(sch)


def test_nr_chain(hiplib):
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(23)
    poly, order = 0x1800063, 24
    for bg, Z, mod, n_cb, rate in ((0, 384, 4, 6, 0.6), (1, 208, 2, 9, 0.35), (0, 64, 3, 12, 0.5)):
        Qm = O.QM[mod]
        N, K = (66 if bg == 0 else 50) * Z, (22 if bg == 0 else 10) * Z
        E = int(K / rate) // Qm * Qm
        h, dem = C.c_void_p(), C.c_void_p()
        capi.check(lib.srsran_hip_nr_sch_create(C.byref(h)), "nr")
        capi.check(lib.srsran_hip_demod_create(C.byref(dem)), "dem")
        msgs = rng.integers(0, 2, (n_cb, K)).astype(np.uint8)
        for m in msgs:
            cs = O.orc().orc_crc_bits(poly, order, O.P(m), K - order)
            m[K - order:] = [(cs >> (order - 1 - j)) & 1 for j in range(order)]
        d_msg, d_cw, d_tx = S.DeviceBuffer.from_numpy(msgs), S.DeviceBuffer(n_cb * N), S.DeviceBuffer(n_cb * E)
        capi.check(lib.srsran_hip_ldpc_encode_batch(h, d_msg.ptr, d_cw.ptr, (capi.HipLdpcCb * n_cb)(*[capi.HipLdpcCb(i * K, i * N, N) for i in range(n_cb)]),
                                                    n_cb, bg, Z, None), "encode")
        capi.check(lib.srsran_hip_ldpc_rm_tx_batch(h, d_cw.ptr, d_tx.ptr, (capi.HipLdpcCb * n_cb)(*[capi.HipLdpcCb(i * N, i * E, E) for i in range(n_cb)]),
                                                   n_cb, bg, Z, 0, mod, N, None), "rm_tx")
        capi.check(lib.srsran_hip_stream_sync(None), "sync")
        tx = d_tx.to_numpy(np.uint8, (n_cb * E,))
        seed = int(rng.integers(0, 1 << 31))
        x = O.modulate(tx ^ O.sequence_bits(seed, n_cb * E), mod)
        sigma = 10 ** (-(6.0 + 5.5 * (mod - 1) + 3.0) / 20) / np.sqrt(2)
        x = (x + sigma * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
        d_x, d_llr = S.DeviceBuffer.from_numpy(x), S.DeviceBuffer(n_cb * E)
        job = (capi.HipDemodJob * 1)(capi.HipDemodJob(mod, x.size, 0, 0, seed, 3))  # demodulate, change sign, descramble
        capi.check(lib.srsran_hip_demod_run(dem, d_x.ptr, d_llr.ptr, capi.LLR_BYTE, job, 1, None), "demod")
        d_soft = S.DeviceBuffer.from_numpy(np.zeros((n_cb, N), np.int8))
        capi.check(lib.srsran_hip_ldpc_rm_rx_batch(h, capi.LLR_BYTE, d_llr.ptr, d_soft.ptr,
                                                   (capi.HipLdpcCb * n_cb)(*[capi.HipLdpcCb(i * E, i * N, E) for i in range(n_cb)]), n_cb, 0, bg, Z, 0, mod, N,
                                                   None), "rm_rx")
        dec = S.LdpcBatch(bg, Z, 0.8, 10, n_cb)
        d_out, d_it = S.DeviceBuffer(n_cb * K), S.DeviceBuffer(n_cb * 4)
        n_llr = min(E, N)
        capi.check(lib.srsran_hip_ldpc_batch_run_crc(dec._h, d_soft.ptr, N, d_out.ptr, K, n_cb, n_llr, poly, order, d_it.ptr, None), "decode")
        capi.check(lib.srsran_hip_stream_sync(None), "sync")
        its, out = d_it.to_numpy(np.int32, (n_cb,)), d_out.to_numpy(np.uint8, (n_cb, K))
        llr = O.sequence_apply((-O.demod_soft(mod, x, "b").astype(np.int32)).astype(np.int8), seed)
        assert np.array_equal(d_llr.to_numpy(np.int8, (n_cb * E,)), llr)
        for i in range(n_cb):
            soft, n2 = O.ldpc_rm_rx(llr[i * E:(i + 1) * E], np.zeros(N, np.int8), 0, bg, Z, 0, mod, N)
            want, rets = O.ldpc_decode(bg, Z, soft[None], 0.8, 10, n2, crc=(poly, order))
            assert its[i] == rets[0] and rets[0] > 0, (bg, Z, i, its[i], rets)
            assert np.array_equal(out[i], want[0]) and np.array_equal(out[i], msgs[i])
        lib.srsran_hip_nr_sch_free(h)
        lib.srsran_hip_demod_free(dem)


def test_nr_transport_block_loopback(hiplib):
    """srsran_hip_sch_nr_encode -> (host: scrambling + constellation mapping + noise) -> demodulate / sign change / descramble ->
    srsran_hip_sch_nr_decode, several transport blocks of different shapes in one call each: payloads come back, verdicts and
    iteration averages equal the oracle loop on the same LLRs (pdsch_nr.c:456-470, sch_nr.c)"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(23)
    #        tbs    R     mod Nl  G
    cases = [(3000, 0.5, 1, 1, 6400), (20040, 0.8, 3, 2, 26400), (50184, 0.75, 4, 1, 67200), (3840, 0.4, 2, 1, 9600)]
    QM = [1, 2, 4, 6, 8]
    cfgs = [O.sch_nr_tb_info(t, R, m, G, Nl, 0) for t, R, m, Nl, G in cases]
    SB, DS = 66 * 384, 8448 // 8
    for c in cfgs:
        c.Nref = c.Z * (66 if c.bg == 0 else 50)
    n_cb = sum(c.C for c in cfgs)
    nr, dem = C.c_void_p(), C.c_void_p()
    capi.check(lib.srsran_hip_sch_nr_create(C.byref(nr), 0.8, 10, n_cb), "sch_nr")
    capi.check(lib.srsran_hip_demod_create(C.byref(dem)), "demod")
    payload = [rng.integers(0, 256, c[0] // 8).astype(np.uint8) for c in cases]
    off_p = np.concatenate([[0], np.cumsum([p.size for p in payload])]).astype(int)
    off_e = np.concatenate([[0], np.cumsum([c[4] for c in cases])]).astype(int)
    first = np.concatenate([[0], np.cumsum([c.C for c in cfgs])]).astype(int)
    tb = (capi.HipNrTb * len(cases))(*[capi.HipNrTb(R, t, m, 0x100, Nl, G, 0, int(off_e[i]), int(off_p[i]), int(first[i]), 0)
                                       for i, (t, R, m, Nl, G) in enumerate(cases)])
    d_pay = S.DeviceBuffer.from_numpy(np.concatenate(payload))
    d_e = S.DeviceBuffer.from_numpy(np.zeros(int(off_e[-1]), np.uint8))
    capi.check(lib.srsran_hip_sch_nr_encode(nr, d_pay.ptr, tb, len(cases), d_e.ptr, None), "encode")
    capi.check(lib.srsran_hip_stream_sync(None), "sync")
    e = d_e.to_numpy(np.uint8, (int(off_e[-1]),))
    syms, seeds, off_s = [], [], [0]
    for i, (t, R, m, Nl, G) in enumerate(cases):
        assert np.array_equal(e[off_e[i]:off_e[i + 1]], O.sch_nr_encode_tb(cfgs[i], 0, payload[i]))
        seeds.append(int(rng.integers(0, 1 << 31)))
        x = O.modulate(e[off_e[i]:off_e[i + 1]] ^ O.sequence_bits(seeds[i], G), m)
        sigma = 10 ** (-(6.0 + 5.0 * m) / 20) / np.sqrt(2)
        syms.append((x + sigma * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64))
        off_s.append(off_s[-1] + x.size)
    d_sym = S.DeviceBuffer.from_numpy(np.concatenate(syms))
    d_llr = S.DeviceBuffer.from_numpy(np.zeros(int(off_e[-1]), np.int8))
    jobs = (capi.HipDemodJob * len(cases))(*[capi.HipDemodJob(cases[i][2], syms[i].size, off_s[i], int(off_e[i]), seeds[i], 3) for i in range(len(cases))])
    capi.check(lib.srsran_hip_demod_run(dem, d_sym.ptr, d_llr.ptr, capi.LLR_BYTE, jobs, len(cases), None), "demod")
    d_soft = S.DeviceBuffer.from_numpy(rng.integers(-50, 50, (n_cb, SB)).astype(np.int8))  # rubbish: the blocks carry the new-data flag
    d_data = S.DeviceBuffer.from_numpy(np.zeros((n_cb, DS), np.uint8))
    d_out = S.DeviceBuffer.from_numpy(np.zeros(int(off_p[-1]), np.uint8))
    cb_crc = np.zeros(n_cb, np.uint8)
    res = (capi.HipNrTbResult * len(cases))()
    capi.check(lib.srsran_hip_sch_nr_decode(nr, d_llr.ptr, tb, len(cases), d_soft.ptr, SB, cb_crc.ctypes.data, d_data.ptr, DS, d_out.ptr, res, None), "decode")
    out = d_out.to_numpy(np.uint8, (int(off_p[-1]),))
    llr = d_llr.to_numpy(np.int8, (int(off_e[-1]),))
    for i, c in enumerate(cfgs):
        assert np.array_equal(llr[off_e[i]:off_e[i + 1]],
                              O.sequence_apply((-O.demod_soft(cases[i][2], syms[i], "b").astype(np.int32)).astype(np.int8), seeds[i])), i
        soft, crc, data = np.zeros((c.C, SB), np.int8), np.zeros(c.C, np.uint8), np.zeros((c.C, DS), np.uint8)
        o, ok, avg = O.sch_nr_decode_tb(c, 0, 0.8, 10, llr[off_e[i]:off_e[i + 1]], soft, crc, data)
        assert (res[i].crc_ok, res[i].nof_cb) == (ok, c.C) and abs(res[i].avg_iter - avg) < 1e-6, i
        assert np.array_equal(cb_crc[first[i]:first[i + 1]], crc), i
        assert ok == 1 and np.array_equal(out[off_p[i]:off_p[i + 1]], payload[i]), i
    lib.srsran_hip_sch_nr_free(nr)
    lib.srsran_hip_demod_free(dem)


def test_empty_batches_are_no_ops(hiplib):
    """every batched entry point accepts an empty batch (as the reference's loops over zero code blocks do nothing)"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    d = S.DeviceBuffer(4096)
    td = S.TdecBatch(512, 4)
    assert lib.srsran_hip_tdec_batch_run(td._h, d.ptr, 3 * 512 + 12, d.ptr, 64, 0, 4, 0, None) == 0
    ld = S.LdpcBatch(0, 8, 0.8, 4, 4)
    assert lib.srsran_hip_ldpc_batch_run(ld._h, d.ptr, 66 * 8, d.ptr, 22 * 8, 0, 66 * 8, None, None) == 0
    of = S.OfdmBatch(6)
    assert lib.srsran_hip_ofdm_batch_rx(of._h, d.ptr, d.ptr, 0, None) == 0
    assert lib.srsran_hip_rm_turbo_rx_batch(d.ptr, 100, 100, d.ptr, 3 * 72 + 12, 0, 40, 0, 0, None) == 0
    for create, free, call in ((lib.srsran_hip_sch_create, lib.srsran_hip_sch_free,
                                lambda h: lib.srsran_hip_sch_decode(h, d.ptr, None, 0, 4, d.ptr, None, d.ptr, None, None)),
                               (lib.srsran_hip_sch_enc_create, lib.srsran_hip_sch_enc_free, lambda h: lib.srsran_hip_sch_encode(h, d.ptr, None, 0, d.ptr, None)),
                               (lib.srsran_hip_demod_create, lib.srsran_hip_demod_free, lambda h: lib.srsran_hip_demod_run(h, d.ptr, d.ptr, 0, None, 0, None)),
                               (lib.srsran_hip_nr_sch_create, lib.srsran_hip_nr_sch_free,
                                lambda h: lib.srsran_hip_ldpc_rm_rx_batch(h, 1, d.ptr, d.ptr, None, 0, 0, 0, 8, 0, 1, 66 * 8, None))):
        h = C.c_void_p()
        assert create(C.byref(h)) == 0
        assert call(h) == 0
        free(h)
    assert lib.srsran_hip_tcod_encode_batch(d.ptr, 40, d.ptr, 132, 0, 40, None) == 0
    assert lib.srsran_hip_predecoding_single(d.ptr, d.ptr, d.ptr, None, 0, 1.0, 0.0, None) == 0
    capi.check(lib.srsran_hip_stream_sync(None), "sync")
