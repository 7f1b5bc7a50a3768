"""Transport-block decoding on the device (decode_tb / decode_tb_cb, sch.c:370-560): rate de-matching into soft buffers,
turbo half iterations with per-code-block CRC early stop, transport-block CRC -- against the oracle's restatement:
identical decoded bytes, CRC verdicts, per-block iteration counts and soft buffers, incl. a HARQ retransmission."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["throughput kernel", "latency kernel", "latency kernel, two waves"])
def tdec_kernel(request, hiplib):
    """every test of this module runs twice: with the throughput kernel (8 code blocks per wave, turbo_kernels.hip) and with the latency kernel
    (one code block per wave, states across lanes, turbo_lat_kernels.hip) wherever the latter exists (16 sub-blocks, and 8 sub-blocks 16-bit) --
    SRSRAN_HIP_TDEC_LAT = 0 / 1; unset, the library picks by batch size (turbo_device.h: kLatMaxBlocks)"""
    # (third run: the two-wave form of the latency kernel -- forward and backward recursion of a block at once, 16 sub-blocks / 16-bit -- SRSRAN_HIP_TDEC_LAT2;
    # the scalar decoder's latency kernel, turbo_gen_lat_kernels.hip, follows SRSRAN_HIP_TDEC_LAT as well)
    assert hiplib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT", b"0" if request.param == "throughput kernel" else b"1") == 0
    assert hiplib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT2", b"1" if request.param == "latency kernel, two waves" else b"0") == 0
    yield request.param
    assert hiplib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT", None) == 0
    assert hiplib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT2", None) == 0
SB = 18600


def _decode(S, capi, lib, h, tb_list, e_all, softbuf, cb_crc, data_len, max_it, d_data=None):
    tbs = (capi.HipTb * len(tb_list))(*tb_list)
    res = (capi.HipTbResult * len(tb_list))()
    d_e = S.DeviceBuffer.from_numpy(e_all)
    d_soft = S.DeviceBuffer.from_numpy(softbuf)
    d_data = d_data or S.DeviceBuffer.from_numpy(np.zeros(data_len, np.uint8))
    capi.check(lib.srsran_hip_sch_decode(h, d_e.ptr, tbs, len(tb_list), max_it, d_soft.ptr, O.P(cb_crc), d_data.ptr, res, None), "sch_decode")
    softbuf[:] = d_soft.to_numpy(np.int16, softbuf.shape)
    return res, d_data.to_numpy(np.uint8, (data_len,)), d_data


def _mask_tail_slots(sb, K):
    """the reference parks the tail LLRs in three unused slots of every stream of the soft buffer when it decodes
    (turbodecoder_iter.h:58-70); the device decoder leaves them alone -- not part of any rate-matching position"""
    out = sb.copy()
    for a in range(3):
        out[a * (K + 32) + K:a * (K + 32) + K + 3] = 0
    return out


def test_batch_of_transport_blocks_vs_oracle(hiplib):
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(3)
    h = C.c_void_p()
    assert lib.srsran_hip_sch_create(C.byref(h)) == 0
    # (tbs, Qm, coded bits G, Es/N0): single block; 13 blocks of 5824 (the C4 grant); 2 x 6144; a noisy one that fails;
    # one at the decoding threshold (different blocks stop at different iterations)
    cases = [(4584, 2, 9000, 3.0), (75376, 6, 100800, 5.0), (12216, 4, 26000, 3.0), (12216, 2, 15000, -6.0), (36696, 6, 52002, 4.5),
             (936, 2, 2400, 3.0)]
    tb_list, e_parts, first_cb, data_off, truth = [], [], 0, 0, []
    for tbs, Qm, G, snr in cases:
        e, payload = O.make_tb(tbs, Qm, G, 0, snr, rng)
        s = O.cbsegm(tbs)
        tb_list.append(capi.HipTb(tbs, Qm, 0, G, sum(p.size for p in e_parts), data_off, first_cb))
        e_parts.append(e)
        truth.append((payload, s))
        first_cb += s["C"]
        data_off += tbs // 8 + 6 + 5  # deliberately unaligned spacing
    e_all = np.concatenate(e_parts)
    softbuf = np.zeros((first_cb, SB), np.int16)
    cb_crc = np.zeros(first_cb, np.uint8)
    res, data, _ = _decode(S, capi, lib, h, tb_list, e_all, softbuf, cb_crc, data_off, 8)
    n_ok = 0
    for i, ((tbs, Qm, G, snr), tb, (payload, s)) in enumerate(zip(cases, tb_list, truth)):
        o_soft = np.zeros((s["C"], SB), np.int16)
        o_crc = np.zeros(s["C"], np.uint8)
        ret, o_data, o_avg = O.sch_decode_tb(tbs, Qm, 0, e_parts[i], o_soft, o_crc, 8)
        assert res[i].crc_ok == ret, (i, res[i].crc_ok, ret)
        assert res[i].nof_cb == s["C"] and abs(res[i].avg_iterations - o_avg) < 1e-6, (i, res[i].avg_iterations, o_avg)
        assert np.array_equal(cb_crc[tb.first_cb:tb.first_cb + s["C"]], o_crc), i
        assert np.array_equal(data[tb.data_offset:tb.data_offset + tbs // 8 + 6], o_data), i
        for c in range(s["C"]):
            K = s["K1"] if c < s["C1"] else s["K2"]
            assert np.array_equal(_mask_tail_slots(softbuf[tb.first_cb + c], K), _mask_tail_slots(o_soft[c], K)), (i, c)
        if ret == 0:
            n_ok += 1
            assert np.array_equal(data[tb.data_offset:tb.data_offset + tbs // 8 + 3], payload)
            assert res[i].avg_iterations < 8  # early stop really happened
    assert n_ok >= 4 and res[3].crc_ok == capi.SRSRAN_ERROR
    lib.srsran_hip_sch_free(h)


def test_new_data_flag_overwrites_the_soft_buffer(hiplib):
    """SRSRAN_HIP_TB_NEW_DATA in rv: the rows of a new block need no srsran_softbuffer_rx_reset -- de-matching overwrites whatever they
    hold (16- and 8-bit), with the result of a cleared buffer"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(21)
    h = C.c_void_p()
    assert lib.srsran_hip_sch_create(C.byref(h)) == 0
    NEW = 0x100
    for dt in (np.int16, np.int8):
        cases = [(75376, 6, 100800, 5.5), (4584, 2, 9000, 3.0), (12216, 4, 26000, 3.5)]
        tb_list, e_parts, first_cb, data_off, segs = [], [], 1, 0, []
        for i, (tbs, Qm, G, snr) in enumerate(cases):
            e16, payload = O.make_tb(tbs, Qm, G, 0, snr, rng)
            e = e16 if dt == np.int16 else np.clip(np.round(e16 * (10.0 / np.mean(np.abs(e16)))), -100, 100).astype(np.int8)
            s = O.cbsegm(tbs)
            tb_list.append(capi.HipTb(tbs, Qm, NEW if i != 1 else 0, G, sum(p.size for p in e_parts), data_off, first_cb))
            e_parts.append(e)
            segs.append(s)
            first_cb += s["C"] + 1
            data_off += tbs // 8 + 8
        lim = 100 if dt == np.int8 else 20000
        softbuf = rng.integers(-lim, lim, (first_cb, SB)).astype(dt)  # rubbish everywhere ...
        for s, tb in zip(segs[1:2], tb_list[1:2]):
            softbuf[tb.first_cb:tb.first_cb + s["C"]] = 0                # ... except under the block without the flag
        guard = softbuf.copy()
        cb_crc = np.zeros(first_cb, np.uint8)
        tbs_arr = (capi.HipTb * len(tb_list))(*tb_list)
        res = (capi.HipTbResult * len(tb_list))()
        d_e = S.DeviceBuffer.from_numpy(np.concatenate(e_parts))
        d_soft = S.DeviceBuffer.from_numpy(softbuf)
        d_data = S.DeviceBuffer.from_numpy(np.zeros(data_off, np.uint8))
        fn = lib.srsran_hip_sch_decode if dt == np.int16 else lib.srsran_hip_sch_decode_8bit
        capi.check(fn(h, d_e.ptr, tbs_arr, len(tb_list), 8, d_soft.ptr, O.P(cb_crc), d_data.ptr, res, None), "decode")
        got = d_soft.to_numpy(dt, softbuf.shape)
        used = np.zeros(first_cb, bool)
        for i, ((tbs, Qm, G, snr), tb, s) in enumerate(zip(cases, tb_list, segs)):
            o_soft, o_crc = np.zeros((s["C"], SB), dt), np.zeros(s["C"], np.uint8)
            ret, o_data, o_avg = O.sch_decode_tb(tbs, Qm, 0, e_parts[i], o_soft, o_crc, 8)
            assert res[i].crc_ok == ret and (ret == 0 or dt == np.int8) and abs(res[i].avg_iterations - o_avg) < 1e-6, (dt, i, ret)
            assert np.array_equal(cb_crc[tb.first_cb:tb.first_cb + s["C"]], o_crc), (dt, i)
            for c in range(s["C"]):
                K = s["K1"] if c < s["C1"] else s["K2"]
                n = 3 * (K + 32) + 12
                assert np.array_equal(_mask_tail_slots(got[tb.first_cb + c][:n], K), _mask_tail_slots(o_soft[c][:n], K)), (dt, i, c)
                used[tb.first_cb + c] = True
        assert np.array_equal(got[~used], guard[~used])  # rows of no block are not touched
    lib.srsran_hip_sch_free(h)


def test_batch_of_transport_blocks_8bit_vs_oracle(hiplib):
    """q->llr_is_8bit (sch.c:408-412,426-428): int8 LLRs, srsran_rm_turbo_rx_lut_8bit, the 8-bit window decoders (32 / 16 / 8
    sub-blocks), same loop; incl. a second transmission of a block whose first one fails"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(8)
    h = C.c_void_p()
    assert lib.srsran_hip_sch_create(C.byref(h)) == 0
    cases = [(4584, 2, 9000, 3.0), (75376, 6, 100800, 5.5), (12216, 4, 26000, 3.5), (12216, 2, 15000, -6.0), (936, 2, 2400, 3.0), (36696, 6, 52002, 3.9)]
    tb_list, e_parts, first_cb, data_off, truth = [], [], 0, 0, []
    for tbs, Qm, G, snr in cases:
        e16, payload = O.make_tb(tbs, Qm, G, 0, snr, rng)
        e = np.clip(np.round(e16 * (10.0 / np.mean(np.abs(e16)))), -100, 100).astype(np.int8)
        s = O.cbsegm(tbs)
        tb_list.append(capi.HipTb(tbs, Qm, 0, G, sum(p.size for p in e_parts), data_off, first_cb))
        e_parts.append(e)
        truth.append((payload, s))
        first_cb += s["C"]
        data_off += tbs // 8 + 6 + 3
    softbuf = np.zeros((first_cb, SB), np.int8)
    cb_crc = np.zeros(first_cb, np.uint8)
    o_soft = [np.zeros((s["C"], SB), np.int8) for _, s in truth]
    o_crc = [np.zeros(s["C"], np.uint8) for _, s in truth]
    o_cbd = [np.zeros((s["C"], 768), np.uint8) for _, s in truth]
    d_data = None
    active = list(range(len(cases)))
    for rnd in range(2):
        if rnd == 1:  # second transmission (rv 2) of the blocks the first one did not recover, fresh noise
            assert active
            for i in active:
                tbs, Qm, G, snr = cases[i]
                e16, _ = O.make_tb(tbs, Qm, G, 2, snr + 8.0, rng, payload=np.unpackbits(truth[i][0][:tbs // 8]))
                e_parts[i] = np.clip(np.round(e16 * (10.0 / np.mean(np.abs(e16)))), -100, 100).astype(np.int8)
                tb_list[i].rv = 2
        e_off = np.cumsum([0] + [p.size for p in e_parts])
        for i in range(len(cases)):
            tb_list[i].e_offset = int(e_off[i])
        sel = [tb_list[i] for i in active]
        tbs_arr = (capi.HipTb * len(sel))(*sel)
        res = (capi.HipTbResult * len(sel))()
        d_e = S.DeviceBuffer.from_numpy(np.concatenate(e_parts))
        d_soft = S.DeviceBuffer.from_numpy(softbuf)
        d_data = d_data or S.DeviceBuffer.from_numpy(np.zeros(data_off, np.uint8))
        capi.check(lib.srsran_hip_sch_decode_8bit(h, d_e.ptr, tbs_arr, len(sel), 8, d_soft.ptr, O.P(cb_crc), d_data.ptr, res, None), "sch_decode_8bit")
        softbuf[:] = d_soft.to_numpy(np.int8, softbuf.shape)
        data = d_data.to_numpy(np.uint8, (data_off,))
        n_ok, failed = 0, []
        for j, i in enumerate(active):
            (tbs, Qm, G, snr), tb, (payload, s) = cases[i], tb_list[i], truth[i]
            ret, o_data, o_avg = O.sch_decode_tb(tbs, Qm, tb.rv, e_parts[i], o_soft[i], o_crc[i], 8, cb_data=o_cbd[i])
            assert res[j].crc_ok == ret, (rnd, i, res[j].crc_ok, ret)
            assert abs(res[j].avg_iterations - o_avg) < 1e-6, (rnd, i, res[j].avg_iterations, o_avg)
            assert np.array_equal(cb_crc[tb.first_cb:tb.first_cb + s["C"]], o_crc[i]), (rnd, i)
            for c in range(s["C"]):
                K = s["K1"] if c < s["C1"] else s["K2"]
                assert np.array_equal(_mask_tail_slots(softbuf[tb.first_cb + c], K), _mask_tail_slots(o_soft[i][c], K)), (rnd, i, c)
            if ret == 0:
                n_ok += 1
                assert np.array_equal(data[tb.data_offset:tb.data_offset + tbs // 8 + 3], payload[:tbs // 8 + 3]), (rnd, i)
            else:
                failed.append(i)
        assert n_ok >= (3 if rnd == 0 else 1), (rnd, n_ok)
        active = failed
    lib.srsran_hip_sch_free(h)


@pytest.mark.parametrize("llr8", [False, True])
def test_mixed_small_and_large_transport_blocks(hiplib, llr8):
    """one call with TBS 16 ... 75,376: blocks of K <= 400 (RAR, paging, SIB, VoLTE sizes) go to the scalar decoder with the natural
    soft-buffer layout exactly as decode_tb_cb does through srsran_tdec_iteration (turbodecoder.c:381-408: gen_impl; the 8-bit API
    widens them to int16, :455-478), side by side with the window decoders -- one small block must not fail the batch"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(21 + llr8)
    h = C.c_void_p()
    assert lib.srsran_hip_sch_create(C.byref(h)) == 0
    # K = tbs + 24: 40, 48, 80, 160, 232, 280, 400 (scalar) | 416, 512 (8 windows) | 4608, 13 x 5824, 2 x 6144; one hopeless small block
    cases = [(16, 2, 120, 4.0), (24, 2, 200, 2.0), (56, 2, 300, 1.0), (75376, 6, 100800, 5.0), (136, 2, 500, 0.0), (208, 4, 640, 1.0),
             (256, 2, 560, 0.5), (376, 2, 900, 0.0), (392, 2, 1000, 1.0), (488, 6, 1200, 1.0), (4584, 2, 9000, 3.0), (12216, 4, 26000, 3.0),
             (328, 2, 720, -8.0), (144, 2, 1800, -4.0)]
    tb_list, e_parts, first_cb, data_off, truth = [], [], 0, 0, []
    for tbs, Qm, G, snr in cases:
        e, payload = O.make_tb(tbs, Qm, G, 0, snr, rng)
        if llr8:
            e = np.clip(np.round(e * (10.0 / np.mean(np.abs(e)))), -100, 100).astype(np.int8)
        s = O.cbsegm(tbs)
        tb_list.append(capi.HipTb(tbs, Qm, 0, G, sum(p.size for p in e_parts), data_off, first_cb))
        e_parts.append(e)
        truth.append((payload, s))
        first_cb += s["C"]
        data_off += tbs // 8 + 6 + 1
    dt = np.int8 if llr8 else np.int16
    softbuf = np.zeros((first_cb, SB), dt)
    cb_crc = np.zeros(first_cb, np.uint8)
    tbs_arr = (capi.HipTb * len(tb_list))(*tb_list)
    res = (capi.HipTbResult * len(tb_list))()
    d_e = S.DeviceBuffer.from_numpy(np.concatenate(e_parts))
    d_soft = S.DeviceBuffer.from_numpy(softbuf)
    d_data = S.DeviceBuffer.from_numpy(np.zeros(data_off, np.uint8))
    f = lib.srsran_hip_sch_decode_8bit if llr8 else lib.srsran_hip_sch_decode
    capi.check(f(h, d_e.ptr, tbs_arr, len(tb_list), 8, d_soft.ptr, O.P(cb_crc), d_data.ptr, res, None), "sch_decode")
    softbuf[:] = d_soft.to_numpy(dt, softbuf.shape)
    data = d_data.to_numpy(np.uint8, (data_off,))
    n_ok = n_small_ok = 0
    for i, ((tbs, Qm, G, snr), tb, (payload, s)) in enumerate(zip(cases, tb_list, truth)):
        o_soft, o_crc = np.zeros((s["C"], SB), dt), np.zeros(s["C"], np.uint8)
        ret, o_data, o_avg = O.sch_decode_tb(tbs, Qm, 0, e_parts[i], o_soft, o_crc, 8)
        assert ret in (0, -1), (i, ret)
        assert res[i].crc_ok == ret, (i, tbs, res[i].crc_ok, ret)
        assert res[i].nof_cb == s["C"] and abs(res[i].avg_iterations - o_avg) < 1e-6, (i, tbs, res[i].avg_iterations, o_avg)
        assert np.array_equal(cb_crc[tb.first_cb:tb.first_cb + s["C"]], o_crc), i
        assert np.array_equal(data[tb.data_offset:tb.data_offset + tbs // 8 + 6], o_data), (i, tbs)
        for c in range(s["C"]):
            K = s["K1"] if c < s["C1"] else s["K2"]
            if K <= 400:  # natural layout: every position is a rate-matching position, nothing is parked in the buffer
                assert np.array_equal(softbuf[tb.first_cb + c], o_soft[c]), (i, c)
            else:
                assert np.array_equal(_mask_tail_slots(softbuf[tb.first_cb + c], K), _mask_tail_slots(o_soft[c], K)), (i, c)
        if ret == 0:
            n_ok += 1
            n_small_ok += tbs <= 376
            assert np.array_equal(data[tb.data_offset:tb.data_offset + tbs // 8 + 3], payload)
    assert n_ok >= 9 and n_small_ok >= 5 and res[12].crc_ok == capi.SRSRAN_ERROR
    assert any(0 < res[i].avg_iterations < 8 for i in range(len(cases)) if cases[i][0] <= 376)  # the scalar decoder stops early too
    lib.srsran_hip_sch_free(h)


def test_harq_retransmission_and_errors(hiplib):
    """first transmission too noisy, the retransmission (rv 2) combines in the soft buffers; code blocks already decoded
    are skipped (their flag is set, their bytes stay in d_data)"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(11)
    h = C.c_void_p()
    assert lib.srsran_hip_sch_create(C.byref(h)) == 0
    tbs, Qm, G = 24496, 4, 30000  # rate 0.82: at this SNR one of the five code blocks decodes in the first round
    s = O.cbsegm(tbs)
    payload_rng = np.random.default_rng(77)
    e0, payload = O.make_tb(tbs, Qm, G, 0, 5.5, np.random.default_rng(5))
    # same payload, redundancy version 2: regenerate with the same payload stream
    e2, payload2 = O.make_tb(tbs, Qm, G, 2, 5.5, np.random.default_rng(5))
    assert np.array_equal(payload, payload2)
    tb = [capi.HipTb(tbs, Qm, 0, G, 0, 0, 0)]
    softbuf, cb_crc = np.zeros((s["C"], SB), np.int16), np.zeros(s["C"], np.uint8)
    o_soft, o_crc, o_keep = softbuf.copy(), cb_crc.copy(), np.zeros((s["C"], 768), np.uint8)
    res, data, d_data = _decode(S, capi, lib, h, tb, e0, softbuf, cb_crc, tbs // 8 + 6, 6)
    ret, o_data, o_avg = O.sch_decode_tb(tbs, Qm, 0, e0, o_soft, o_crc, 6, o_keep)
    assert res[0].crc_ok == ret == capi.SRSRAN_ERROR and np.array_equal(cb_crc, o_crc) and abs(res[0].avg_iterations - o_avg) < 1e-6
    assert 0 < cb_crc.sum() < s["C"] and np.array_equal(data, o_data)
    tb = [capi.HipTb(tbs, Qm, 2, G, 0, 0, 0)]
    res, data, _ = _decode(S, capi, lib, h, tb, e2, softbuf, cb_crc, tbs // 8 + 6, 6, d_data)
    # oracle: second round on its own buffers; blocks decoded in round 1 keep their bytes
    ret2, o_data2, o_avg2 = O.sch_decode_tb(tbs, Qm, 2, e2, o_soft, o_crc, 6, o_keep)
    assert ret2 == 0 and np.array_equal(data, o_data2)
    for c in range(s["C"]):
        K = s["K1"] if c < s["C1"] else s["K2"]
        assert np.array_equal(_mask_tail_slots(softbuf[c], K), _mask_tail_slots(o_soft[c], K))
    assert res[0].crc_ok == capi.SRSRAN_SUCCESS and np.all(cb_crc == 1)
    assert np.array_equal(data[:tbs // 8 + 3], payload)
    assert abs(res[0].avg_iterations - o_avg2) < 1e-6
    # a third call finds every code block decoded: no decoder is launched, the transport-block CRC runs over the bytes that are still in
    # d_data and passes again, no iteration is counted.  (Not a case the reference is ever in -- a decoded block's soft buffer is reset --
    # and not comparable with it: sch.c:478-485 saves a block's good code blocks for the next round only when the block FAILED.)
    res, data3, _ = _decode(S, capi, lib, h, tb, e2, softbuf, cb_crc, tbs // 8 + 6, 6, d_data)
    assert res[0].crc_ok == capi.SRSRAN_SUCCESS and res[0].avg_iterations == 0.0 and res[0].nof_cb == s["C"]
    assert np.array_equal(data3, data) and np.all(cb_crc == 1)
    # errors: filler bits (non-standard TBS), bad arguments
    d = S.DeviceBuffer(1 << 20)
    r = (capi.HipTbResult * 1)()
    flags = np.zeros(4, np.uint8)
    for bad in (capi.HipTb(6208, 2, 0, 20000, 0, 0, 0), capi.HipTb(4584, 0, 0, 9000, 0, 0, 0), capi.HipTb(4584, 2, 4, 9000, 0, 0, 0)):
        flags[:] = 0
        assert lib.srsran_hip_sch_decode(h, d.ptr, (capi.HipTb * 1)(bad), 1, 8, d.ptr, O.P(flags), d.ptr, r, None) == capi.SRSRAN_ERROR_INVALID_INPUTS
    lib.srsran_hip_sch_free(h)


class _SoftbufferRx(C.Structure):  # srsran_softbuffer_rx_t, softbuffer.h:40-47
    _fields_ = [("max_cb", C.c_uint32), ("max_cb_size", C.c_uint32), ("buffer_f", C.POINTER(C.c_void_p)), ("data", C.POINTER(C.c_void_p)),
                ("cb_crc", C.POINTER(C.c_bool)), ("tb_crc", C.c_bool)]


class _SchHead(C.Structure):  # srsran_sch_t up to llr_is_8bit, sch.h:51-57 (the reference object is 0.9 MB behind it: never touched)
    _fields_ = [("max_iterations", C.c_uint32), ("avg_iterations", C.c_float), ("llr_is_8bit", C.c_bool), ("guard", C.c_uint8 * 64)]


def _host_softbuffer(max_cb, dt):
    """what srsran_softbuffer_rx_init_guru allocates and srsran_softbuffer_rx_reset clears: one row of 18600 int16 and 18600 / 8 bytes per block"""
    rows = [np.zeros(SB, np.int16) for _ in range(max_cb)]
    keep = [np.zeros(SB // 8, np.uint8) for _ in range(max_cb)]
    flags = np.zeros(max_cb, np.bool_)
    sb = _SoftbufferRx(max_cb, SB, (C.c_void_p * max_cb)(*[r.ctypes.data for r in rows]), (C.c_void_p * max_cb)(*[k.ctypes.data for k in keep]),
                       flags.ctypes.data_as(C.POINTER(C.c_bool)), False)
    return sb, rows, keep, flags


@pytest.mark.parametrize("llr8", [False, True], ids=["16bit", "8bit"])
def test_decode_tb_cb_on_the_reference_structs(hiplib, llr8):
    """the reference's own seam, decode_tb_cb (sch.c:370-492), on HOST buffers in the reference's structs: srsran_softbuffer_rx_t rows, flags and
    stored blocks, q->max_iterations / avg_iterations / llr_is_8bit.  A noisy first transmission (some code blocks decode), the retransmission
    with rv 2 combining in the rows that came back, and the 13-block grant of the uplink configuration -- equal to the oracle's decode_tb_cb in
    return value, bytes, flags, stored blocks, iteration average and the soft rows of the blocks that are still undecoded"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    fn = lib.srsran_hip_decode_tb_cb
    fn.restype = C.c_bool
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    dt = np.int8 if llr8 else np.int16

    def q8(e16):
        return e16 if not llr8 else np.clip(np.round(e16 * (10.0 / np.mean(np.abs(e16)))), -100, 100).astype(np.int8)

    def seg_struct(tbs):
        cs = capi.Cbsegm()
        assert lib.srsran_cbsegm(C.byref(cs), tbs) == 0
        return cs

    def views(rows):
        return [r.view(np.int8)[:SB] if llr8 else r for r in rows]

    q = _SchHead(6, -1.0, llr8)
    q.guard[:] = [0xA5] * 64
    # ---- HARQ: rate 0.82 at an SNR where some of the five code blocks decode in the first round
    tbs, Qm, G = 24496, 4, 30000
    s = O.cbsegm(tbs)
    snr = 5.5 if not llr8 else 6.2  # (found with the oracle: two of the five blocks fail the first round at either width)
    e0, payload = O.make_tb(tbs, Qm, G, 0, snr, np.random.default_rng(5))
    e2, _ = O.make_tb(tbs, Qm, G, 2, snr, np.random.default_rng(5))
    e0, e2 = q8(e0), q8(e2)
    sb, rows, keep, flags = _host_softbuffer(s["C"] + 2, dt)
    o_soft, o_crc, o_keep = np.zeros((s["C"], SB), dt), np.zeros(s["C"], np.uint8), np.zeros((s["C"], 768), np.uint8)
    cs = seg_struct(tbs)
    data = np.full(tbs // 8 + 6 + 16, 0xEE, np.uint8)
    ok = fn(C.byref(q), C.byref(sb), C.byref(cs), Qm, 0, G, O.P(e0), O.P(data))
    ret, o_data, o_avg = O.sch_decode_tb(tbs, Qm, 0, e0, o_soft, o_crc, 6, o_keep)
    assert (not ok) and ret == -1 and not sb.tb_crc
    assert np.array_equal(flags[:s["C"]].astype(np.uint8), o_crc) and 0 < o_crc.sum() < s["C"], o_crc
    assert abs(q.avg_iterations - o_avg) < 1e-6 and np.array_equal(data[:tbs // 8 + 6], o_data) and np.all(data[tbs // 8 + 6:] == 0xEE)
    for c in range(s["C"]):
        K = s["K1"] if c < s["C1"] else s["K2"]
        rl = (K - 24) // 8
        if o_crc[c]:
            assert np.array_equal(keep[c][:rl], o_keep[c][:rl]), c  # a decoded block's bytes are stored for the next round (sch.c:476-484)
        else:
            assert np.array_equal(_mask_tail_slots(views(rows)[c], K), _mask_tail_slots(o_soft[c], K)), c  # ... an undecoded one's soft bits combined
    # second transmission: the rows as the first call left them on the HOST, the stored blocks copied into `data`
    data2 = np.full(tbs // 8 + 6, 0x11, np.uint8)
    ok = fn(C.byref(q), C.byref(sb), C.byref(cs), Qm, 2, G, O.P(e2), O.P(data2))
    ret2, o_data2, o_avg2 = O.sch_decode_tb(tbs, Qm, 2, e2, o_soft, o_crc, 6, o_keep)
    assert ok and ret2 == 0 and sb.tb_crc and np.all(flags[:s["C"]]) and abs(q.avg_iterations - o_avg2) < 1e-6
    assert np.array_equal(data2[:tbs // 8 + 3], payload) and np.array_equal(data2[:tbs // 8 + 3], o_data2[:tbs // 8 + 3])
    assert bytes(q.guard) == bytes([0xA5] * 64)
    # every code block decoded already (the reference's callers reset such a buffer first; its loop would only copy the stored blocks, :466-471)
    # -- decode_tb_cb then copies the STORED blocks, which were only kept while the block failed (:476): same bytes as the oracle's loop
    data3 = np.full(tbs // 8 + 6, 0x22, np.uint8)
    assert fn(C.byref(q), C.byref(sb), C.byref(cs), Qm, 2, G, O.P(e2), O.P(data3)) and sb.tb_crc and q.avg_iterations == 0.0
    _, o_data3, _ = O.sch_decode_tb(tbs, Qm, 2, e2, o_soft, o_crc, 6, o_keep)
    rl = [((s["K1"] if c < s["C1"] else s["K2"]) - 24) // 8 for c in range(s["C"])]
    for c in range(s["C"]):
        assert np.array_equal(data3[sum(rl[:c]):sum(rl[:c + 1])], o_data3[sum(rl[:c]):sum(rl[:c + 1])]), c
    # ---- srsran_softbuffer_rx_reset (rows and flags zero), then the 13-block grant of the uplink configuration, first transmission decodes
    for r in rows:
        r[:] = 0
    flags[:] = False
    q.max_iterations = 8
    tbs, Qm, G = 75376, 6, 100800
    s = O.cbsegm(tbs)
    sb, rows, keep, flags = _host_softbuffer(s["C"], dt)
    e, payload = O.make_tb(tbs, Qm, G, 0, 6.0, np.random.default_rng(9))
    e = q8(e)
    cs = seg_struct(tbs)
    data = np.zeros(tbs // 8 + 6, np.uint8)
    ok = fn(C.byref(q), C.byref(sb), C.byref(cs), Qm, 0, G, O.P(e), O.P(data))
    o_soft, o_crc = np.zeros((s["C"], SB), dt), np.zeros(s["C"], np.uint8)
    ret, o_data, o_avg = O.sch_decode_tb(tbs, Qm, 0, e, o_soft, o_crc, 8)
    assert ok and ret == 0 and np.all(flags) and abs(q.avg_iterations - o_avg) < 1e-6 and q.avg_iterations < 8
    assert np.array_equal(data, o_data) and np.array_equal(data[:tbs // 8 + 3], payload)
    # ---- single-block transport blocks: CRC24A on the block itself (sch.c:432-438); K <= 400 goes to the scalar decoder (turbodecoder.c:381-408)
    for tbs, Qm, G, snr_db in ((4584, 2, 9000, 3.0), (936, 2, 2400, 3.0), (152, 2, 600, 4.0), (16, 2, 144, 5.0), (4584, 2, 5200, -2.0)):
        s = O.cbsegm(tbs)
        assert s["C"] == 1
        sb, rows, keep, flags = _host_softbuffer(2, dt)
        e, payload = O.make_tb(tbs, Qm, G, 0, snr_db, np.random.default_rng(tbs))
        e = q8(e)
        cs = seg_struct(tbs)
        data = np.zeros(tbs // 8 + 6, np.uint8)
        q.max_iterations = 5
        ok = fn(C.byref(q), C.byref(sb), C.byref(cs), Qm, 0, G, O.P(e), O.P(data))
        o_soft, o_crc = np.zeros((1, SB), dt), np.zeros(1, np.uint8)
        ret, o_data, o_avg = O.sch_decode_tb(tbs, Qm, 0, e, o_soft, o_crc, 5)
        assert bool(ok) == bool(o_crc[0]) and bool(flags[0]) == bool(o_crc[0]) and abs(q.avg_iterations - o_avg) < 1e-6, (tbs, G, ok, o_crc, q.avg_iterations, o_avg)
        K = s["K1"]
        assert np.array_equal(data[:K // 8], o_data[:K // 8]), (tbs, G)
        if not o_crc[0]:
            n = 3 * (K + 32) + 12 if K > 400 else 3 * K + 12
            got, want = views(rows)[0][:n], o_soft[0][:n]
            assert np.array_equal(_mask_tail_slots(got, K) if K > 400 else got, _mask_tail_slots(want, K) if K > 400 else want), (tbs, G)
    # bad arguments fail loudly, nothing is written
    assert not fn(None, C.byref(sb), C.byref(cs), Qm, 0, G, O.P(e), O.P(data))
    assert not fn(C.byref(q), C.byref(sb), C.byref(cs), 0, 0, G, O.P(e), O.P(data))


def test_turbo_encoder(hiplib):
    """srsran_tcod_encode (turbocoder.c:76-185) drop-in and batched, every block size, filler marks"""
    import ctypes as C

    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(12)
    q = capi.Tcod()
    assert lib.srsran_tcod_init(C.byref(q), 6144) == 0
    orc = O.orc()
    for K in (40, 48, 512, 1008, 2112, 6144):
        x = rng.integers(0, 2, K).astype(np.uint8)
        x[:min(K // 4, 28)] = 100  # SRSRAN_TX_NULL
        out, want = np.zeros(3 * K + 12, np.uint8), np.zeros(3 * K + 12, np.uint8)
        assert lib.srsran_tcod_encode(C.byref(q), O.P(x), O.P(out), K) == 0
        assert orc.orc_tcod_encode(O.P(x), O.P(want), K) == 0
        assert np.array_equal(out, want), K
    x = np.zeros(41, np.uint8)
    assert lib.srsran_tcod_encode(C.byref(q), O.P(x), O.P(out), 41) == -1  # "Invalid CB size"
    small = capi.Tcod()
    assert lib.srsran_tcod_init(C.byref(small), 512) == 0
    assert lib.srsran_tcod_encode(C.byref(small), O.P(x), O.P(out), 1024) == -1  # initiated for max_long_cb=512
    lib.srsran_tcod_free(C.byref(small))
    lib.srsran_tcod_free(C.byref(q))
    for K in O.tc_sizes():
        n_cb = 3
        x = rng.integers(0, 2, (n_cb, K)).astype(np.uint8)
        d_in, d_out = S.DeviceBuffer.from_numpy(x), S.DeviceBuffer(n_cb * (3 * K + 12))
        capi.check(lib.srsran_hip_tcod_encode_batch(d_in.ptr, K, d_out.ptr, 3 * K + 12, n_cb, K, None), "tcod batch")
        capi.check(lib.srsran_hip_stream_sync(None), "sync")
        got = d_out.to_numpy(np.uint8, (n_cb, 3 * K + 12))
        for i in range(n_cb):
            assert np.array_equal(got[i], O.turbo_encode(x[i])), (K, i)


def test_transport_block_encode(hiplib):
    """srsran_hip_sch_encode == encode_tb_off of the reference (fixture built from its CRC / tcod_encode_lut / rm_turbo_tx_lut),
    several transport blocks per call at unaligned bit offsets, and a device loop-back through srsran_hip_sch_decode"""
    import ctypes as C
    import os

    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sch_tx_ref.npz"))
    h = C.c_void_p()
    capi.check(lib.srsran_hip_sch_enc_create(C.byref(h)), "enc_create")
    keys = [str(k) for k in d["cases"]]
    datas, tb, want, off_b, off_e = [], [], [], 0, 5
    for key in keys:
        tbs, Qm, rv, nof_e = [int(t.lstrip("tbqrvg")) for t in key.split("_")]
        datas.append(d[key + "_data"])
        tb.append(capi.HipTb(tbs, Qm, rv, nof_e, off_e, off_b, 0))
        want.append((off_e, nof_e, np.unpackbits(d[key + "_e"])[:nof_e // Qm * Qm]))
        off_b += tbs // 8
        off_e += nof_e + 3
    d_data = S.DeviceBuffer.from_numpy(np.concatenate(datas))
    d_e = S.DeviceBuffer.from_numpy(np.full(off_e // 8 + 16, 0xFF, np.uint8))
    arr = (capi.HipTb * len(tb))(*tb)
    for _ in range(2):
        capi.check(lib.srsran_hip_sch_encode(h, d_data.ptr, arr, len(tb), d_e.ptr, None), "sch_encode")
    capi.check(lib.srsran_hip_stream_sync(None), "sync")
    got = np.unpackbits(d_e.to_numpy(np.uint8, (off_e // 8 + 16,)))
    for key, (o, n, e) in zip(keys, want):
        assert np.array_equal(got[o:o + e.size], e), key
    bad = (capi.HipTb * 1)(capi.HipTb(6208, 2, 0, 9000, 0, 0, 0))  # filler bits
    assert lib.srsran_hip_sch_encode(h, d_data.ptr, bad, 1, d_e.ptr, None) == capi.SRSRAN_ERROR_INVALID_INPUTS
    # loop-back: 24 transport blocks encoded on the device, BPSK soft bits, decoded on the device
    rng = np.random.default_rng(3)
    tbs, Qm, G, n_tb = 75376, 6, 100800, 24
    ncb = O.cbsegm(tbs)["C"]
    payload = rng.integers(0, 256, (n_tb, tbs // 8)).astype(np.uint8)
    d_data = S.DeviceBuffer.from_numpy(payload)
    d_e = S.DeviceBuffer(n_tb * G // 8)
    tbv = (capi.HipTb * n_tb)(*[capi.HipTb(tbs, Qm, 0, G, i * G, i * (tbs // 8), i * ncb) for i in range(n_tb)])
    capi.check(lib.srsran_hip_sch_encode(h, d_data.ptr, tbv, n_tb, d_e.ptr, None), "sch_encode")
    capi.check(lib.srsran_hip_stream_sync(None), "sync")
    e = np.unpackbits(d_e.to_numpy(np.uint8, (n_tb * G // 8,))).reshape(n_tb, G)
    assert np.array_equal(e[0], O.tb_coded_bits(tbs, Qm, G, 0, None, payload=np.unpackbits(payload[0]), tx_order=True)[0])
    llr = np.clip(np.round(40.0 * ((2.0 * e - 1.0) + 0.45 * rng.standard_normal(e.shape))), -32768, 32767).astype(np.int16)
    d_llr = S.DeviceBuffer.from_numpy(llr)
    dlen = tbs // 8 + 8
    d_out = S.DeviceBuffer.from_numpy(np.zeros((n_tb, dlen), np.uint8))
    d_soft = S.DeviceBuffer.from_numpy(np.zeros((n_tb * ncb, capi.SOFTBUFFER_CB_SIZE), np.int16))
    flags = np.zeros(n_tb * ncb, np.uint8)
    res = (capi.HipTbResult * n_tb)()
    rx = (capi.HipTb * n_tb)(*[capi.HipTb(tbs, Qm, 0, G, i * G, i * dlen, i * ncb) for i in range(n_tb)])
    hd = C.c_void_p()
    capi.check(lib.srsran_hip_sch_create(C.byref(hd)), "sch_create")
    capi.check(lib.srsran_hip_sch_decode(hd, d_llr.ptr, rx, n_tb, 8, d_soft.ptr, flags.ctypes.data, d_out.ptr, res, None), "sch_decode")
    out = d_out.to_numpy(np.uint8, (n_tb, dlen))
    assert all(r.crc_ok == 0 for r in res)
    assert np.array_equal(out[:, :tbs // 8], payload)
    lib.srsran_hip_sch_free(hd)
    lib.srsran_hip_sch_enc_free(h)


@pytest.mark.parametrize("llr8", [False, True], ids=["16bit", "8bit"])
def test_decode_tb_cb_random_harq_sequences(hiplib, llr8):
    """the seam over randomly drawn HARQ processes (fixed seed): transport block sizes of 1 ... 6 code blocks, Qm 2 / 4 / 6, rates around and above
    what the first transmission can carry, up to four transmissions in the order rv 0, 2, 3, 1 with srsran_softbuffer_rx_reset in front of the
    first -- after EVERY call the return value, flags, stored blocks, bytes, iteration average and the rows of the blocks that are still undecoded
    equal the oracle's decode_tb_cb run on its own buffers"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    fn = lib.srsran_hip_decode_tb_cb
    fn.restype = C.c_bool
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    dt = np.int8 if llr8 else np.int16
    rng = np.random.default_rng(2024 + int(llr8))
    sizes = [152, 936, 2216, 4584, 6200, 9144, 15264, 24496, 36696]
    n_calls = n_fail = n_retx_ok = 0
    for trial in range(14):
        tbs = int(sizes[rng.integers(0, len(sizes))])
        Qm = int(rng.choice([2, 4, 6]))
        rate = float(rng.uniform(0.55, 1.05))
        G = max(Qm * 24, int(tbs / rate) // Qm * Qm)
        snr = float(rng.uniform(-1.0, 5.0)) + (2.0 if rate > 0.85 else 0.0)
        s = O.cbsegm(tbs)
        Cn = s["C"]
        q = _SchHead(int(rng.integers(2, 7)), -1.0, llr8)
        sb, rows, keep, flags = _host_softbuffer(Cn + 1, dt)
        o_soft, o_crc, o_keep = np.zeros((Cn, SB), dt), np.zeros(Cn, np.uint8), np.zeros((Cn, 768), np.uint8)
        cs = capi.Cbsegm()
        assert lib.srsran_cbsegm(C.byref(cs), tbs) == 0
        payload_bits = None
        for rv in (0, 2, 3, 1):
            e16, payload = O.make_tb(tbs, Qm, G, rv, snr, rng, payload=payload_bits)
            payload_bits = np.unpackbits(payload)[:tbs]
            e = e16 if not llr8 else np.clip(np.round(e16 * (10.0 / max(1.0, np.mean(np.abs(e16))))), -100, 100).astype(np.int8)
            data = np.full(tbs // 8 + 6, 0x5A, np.uint8)
            ok = fn(C.byref(q), C.byref(sb), C.byref(cs), Qm, rv, G, O.P(e), O.P(data))
            ret, o_data, o_avg = O.sch_decode_tb(tbs, Qm, rv, e, o_soft, o_crc, q.max_iterations, o_keep)
            n_calls += 1
            tag = (trial, tbs, Qm, G, rv, round(snr, 2))
            assert bool(ok) == bool(o_crc.all()) and np.array_equal(flags[:Cn].astype(np.uint8), o_crc) and not flags[Cn], tag
            assert abs(q.avg_iterations - o_avg) < 1e-6, (tag, q.avg_iterations, o_avg)
            for c in range(Cn):
                K = s["K1"] if c < s["C1"] else s["K2"]
                rl = (K if Cn == 1 else K - 24) // 8
                at = sum(((s["K1"] if j < s["C1"] else s["K2"]) - (0 if Cn == 1 else 24)) // 8 for j in range(c))
                assert np.array_equal(data[at:at + rl], o_data[at:at + rl]), (tag, c)
                if o_crc[c]:
                    if not o_crc.all():
                        assert np.array_equal(keep[c][:rl], o_keep[c][:rl]), (tag, c)
                else:
                    n = 3 * (K + 32) + 12 if K > 400 else 3 * K + 12
                    got = (rows[c].view(np.int8)[:SB] if llr8 else rows[c])[:n]
                    want = o_soft[c][:n]
                    assert np.array_equal(_mask_tail_slots(got, K) if K > 400 else got, _mask_tail_slots(want, K) if K > 400 else want), (tag, c)
            if ok:
                assert ret == 0 and np.array_equal(data[:tbs // 8], payload[:tbs // 8]), tag
                n_retx_ok += int(rv != 0)
                break
            n_fail += 1
    assert n_fail >= 4 and n_retx_ok >= 2, (n_calls, n_fail, n_retx_ok)  # the draw does exercise failures and combining


class _SoftbufferTx(C.Structure):  # srsran_softbuffer_tx_t, softbuffer.h:49-53
    _fields_ = [("max_cb", C.c_uint32), ("max_cb_size", C.c_uint32), ("buffer_b", C.POINTER(C.c_void_p))]


def test_encode_tb_on_the_reference_structs(hiplib):
    """srsran_hip_encode_tb = encode_tb (sch.c:239-368) as srsran_dlsch_encode2 reaches it, on host buffers: every recorded reference case
    (its CRC / srsran_tcod_encode_lut / srsran_rm_turbo_tx_lut chain: all redundancy versions, Qm 2 / 4 / 6, one and several code blocks), the
    bits behind the block's end in the last byte left alone or cleared as the reference does, and a retransmission with data == NULL from
    what the first call left in the soft buffer rows"""
    import os

    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    fn = lib.srsran_hip_encode_tb
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sch_tx_ref.npz"))
    rows = [np.zeros(SB, np.uint8) for _ in range(14)]
    sb = _SoftbufferTx(14, SB, (C.c_void_p * 14)(*[r.ctypes.data for r in rows]))
    seen = {}
    for key in [str(k) for k in d["cases"]]:
        tbs, Qm, rv, nof_e = [int(t.lstrip("tbqrvg")) for t in key.split("_")]
        cs = capi.Cbsegm()
        assert lib.srsran_cbsegm(C.byref(cs), tbs) == 0
        data = np.ascontiguousarray(d[key + "_data"])
        want = np.unpackbits(d[key + "_e"])[:nof_e // Qm * Qm]
        out = np.full(nof_e // 8 + 9, 0xFF, np.uint8)
        assert fn(C.byref(sb), C.byref(cs), Qm, rv, nof_e, O.P(data), O.P(out)) == 0, key
        got = np.unpackbits(out)
        assert np.array_equal(got[:want.size], want), key
        assert np.all(out[(nof_e + 7) // 8:] == 0xFF), key  # nothing is written behind the last byte
        # data == NULL: the same block again, another redundancy version, from the rows
        rv2 = (rv + 2) % 4
        k2 = "tbs%d_q%d_rv%d_g%d" % (tbs, Qm, rv2, nof_e)
        out2 = np.full(nof_e // 8 + 9, 0xFF, np.uint8)
        assert fn(C.byref(sb), C.byref(cs), Qm, rv2, nof_e, None, O.P(out2)) == 0
        if k2 + "_e" in d.files and np.array_equal(d[k2 + "_data"], data):
            w2 = np.unpackbits(d[k2 + "_e"])[:nof_e // Qm * Qm]
            assert np.array_equal(np.unpackbits(out2)[:w2.size], w2), k2
            seen[k2] = True
        else:  # no recording of that version for this payload: the oracle's chain
            e_or = O.tb_coded_bits(tbs, Qm, nof_e, rv2, None, payload=np.unpackbits(data), tx_order=True)[0]
            assert np.array_equal(np.unpackbits(out2)[:e_or.size], e_or), (key, rv2)
    # protection (sch.c:249-267,351)
    cs = capi.Cbsegm()
    assert lib.srsran_cbsegm(C.byref(cs), 4584) == 0
    assert fn(C.byref(sb), C.byref(cs), 0, 0, 9000, O.P(data), O.P(out)) < 0
    assert fn(None, C.byref(cs), 2, 0, 9000, O.P(data), O.P(out)) < 0
    sb.max_cb = 0
    assert fn(C.byref(sb), C.byref(cs), 2, 0, 9000, O.P(data), O.P(out)) < 0


def test_byte_packed_encoder_and_rate_matcher(hiplib):
    """srsran_tcod_encode_lut / srsran_rm_turbo_tx_lut (turbocoder.c:188-343, rm_turbo.c:340-378) against the reference's recorded
    outputs: CRC bytes and tail nibble written into `input`, parity bytes, the running transport-block checksum, and the
    rate-matched bits at several bit offsets / lengths / redundancy versions including the bits the copy leaves untouched or clears"""
    import ctypes as C
    import os

    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tcod_lut_ref.npz"))
    tc = capi.Tcod()
    assert lib.srsran_tcod_init(C.byref(tc), 6144) == 0
    lib.srsran_tcod_gentable()
    for key in d["cases"]:
        key = str(key)
        idx, K, with_cb, last, state0, state1, ret = [int(v) for v in d[key + "_meta"]]
        crc_tb, crc_cb = capi.Crc(), capi.Crc()
        for c, poly in ((crc_tb, 0x1864CFB), (crc_cb, 0x1800063)):
            c.polynom, c.order, c.crcmask, c.crchighbit = poly, 24, 0xFFFFFF, 1 << 23
        crc_tb.crcinit = state0
        inp = np.zeros(K // 8 + 8, np.uint8)
        inp[:d[key + "_in"].size] = d[key + "_in"]
        par = np.zeros(K // 4 + 8, np.uint8)
        got = lib.srsran_tcod_encode_lut(C.byref(tc), C.byref(crc_tb), C.byref(crc_cb) if with_cb else None, O.P(inp), O.P(par), idx, bool(last))
        assert got == ret == 3 * K + 12
        assert np.array_equal(inp[:K // 8 + 1], d[key + "_sys"]), key
        assert np.array_equal(par[:K // 4 + 1], d[key + "_par"]), key
        assert (crc_tb.crcinit & 0xFFFFFF) == state1, key
        w_buff = np.zeros(3 * 6176, np.uint8)
        at = 0
        for rv, out_len, w_off, size in d[key + "_txpar"]:
            o = np.full(int(size), 0xA5, np.uint8)
            assert lib.srsran_rm_turbo_tx_lut(O.P(w_buff), O.P(inp), O.P(par), O.P(o), idx, int(out_len), int(w_off), int(rv)) == 0
            assert np.array_equal(o, d[key + "_tx"][at:at + int(size)]), (key, int(rv), int(out_len), int(w_off))
            at += int(size)
    assert lib.srsran_tcod_encode_lut(C.byref(tc), C.byref(crc_tb), None, O.P(inp), O.P(par), 188, False) == -1
    assert lib.srsran_rm_turbo_tx_lut(O.P(w_buff), O.P(inp), O.P(par), O.P(o), 0, 8, 0, 4) == capi.SRSRAN_ERROR_INVALID_INPUTS
    lib.srsran_tcod_free(C.byref(tc))
