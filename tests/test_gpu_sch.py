"""Transport-block decoding on the device (decode_tb / decode_tb_cb, sch.c:370-560): rate de-matching into soft buffers,
turbo half iterations with per-code-block CRC early stop, transport-block CRC -- against the oracle's restatement:
identical decoded bytes, CRC verdicts, per-block iteration counts and soft buffers, incl. a HARQ retransmission."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu
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
    # errors: filler bits (non-standard TBS), scalar-decoder block sizes, bad arguments
    d = S.DeviceBuffer(1 << 20)
    r = (capi.HipTbResult * 1)()
    flags = np.zeros(4, np.uint8)
    for bad in (capi.HipTb(6208, 2, 0, 20000, 0, 0, 0), capi.HipTb(256, 2, 0, 1000, 0, 0, 0), capi.HipTb(4584, 0, 0, 9000, 0, 0, 0),
                capi.HipTb(4584, 2, 4, 9000, 0, 0, 0)):
        flags[:] = 0
        assert lib.srsran_hip_sch_decode(h, d.ptr, (capi.HipTb * 1)(bad), 1, 8, d.ptr, O.P(flags), d.ptr, r, None) == capi.SRSRAN_ERROR_INVALID_INPUTS
    lib.srsran_hip_sch_free(h)
