import sys, ctypes as C, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_api as O
import srslte_amd as S
from srslte_amd import capi
lib = S.lib()
prb, N = 6, 128
flen = 15 * N
q = capi.Sync()
assert lib.srsran_sync_init(C.byref(q), flen, flen, N) == 0
lib.srsran_sync_set_threshold(C.byref(q), 5.0)
lib.srsran_sync_set_em_alpha(C.byref(q), 1.0)
lib.srsran_sync_set_sss_algorithm(C.byref(q), capi.SSS_PARTIAL_3)
for cid in (0, 1, 2):
    lib.srsran_sync_set_N_id_2(C.byref(q), cid % 3)
    sig, end = O.cell_signal(cid, prb, N, False, sf5=False, n_sf=1)
    off = 100 + cid % 17
    buf = np.zeros(2 * flen, np.complex64); buf[off:off + flen] = sig
    pk = C.c_uint32()
    ret = lib.srsran_sync_find(C.byref(q), O.P(buf), 0, C.byref(pk))
    print("cid", cid, "ret", ret, "pk", pk.value, "want", off + end, "peak_value", q.peak_value, "pss.peak", q.pss.peak_value, "oracle", O.pss_find(buf[:flen], N, cid % 3))
    pv = C.c_float()
    p = lib.srsran_pss_find_pss(C.byref(q.pss), O.P(buf), C.byref(pv))
    print("  direct find_pss:", p, pv.value, q.pss.peak_value, q.pss.frame_size, q.pss.fft_size, q.pss.ema_alpha)
