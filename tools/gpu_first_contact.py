# first GPU contact: small parity runs of the three kernels against the oracle
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle_api as O
import srslte_amd as S
from srslte_amd import capi
print(S.lib().srsran_hip_build_info().decode(), "devices:", S.lib().srsran_hip_device_count(), flush=True)
ok_all = True
# ---- turbo
for K, impl_o, impl_g in [(6144, O.ORC_TDEC_AUTO, capi.TDEC_AUTO), (1024, O.ORC_TDEC_AUTO, capi.TDEC_AUTO), (512, O.ORC_TDEC_AUTO, capi.TDEC_AUTO),
                          (816, O.ORC_TDEC_AUTO, capi.TDEC_AUTO), (40, O.ORC_TDEC_AUTO, capi.TDEC_AUTO), (400, O.ORC_TDEC_AUTO, capi.TDEC_AUTO), (2112, O.ORC_TDEC_SSE_WINDOW, capi.TDEC_SSE_WINDOW)]:
    for snr in [3.0, -1.0, -4.0]:
        n_cb = 19
        msgs, llr = O.turbo_llrs(K, n_cb, snr, seed=K + int(snr * 10))
        dec = S.TdecBatch(K, n_cb, impl_g)
        for nit in [1, 2, 3, 8]:
            t0 = time.time()
            ref, ref_llr = O.turbo_decode(llr, nit, K, impl_o, want_llr=True)
            out, out_llr = dec.decode(llr, nit, want_llr=True)
            okb = np.array_equal(ref, out); okl = np.array_equal(ref_llr, out_llr)
            ok_all &= okb and okl
            if not (okb and okl) or nit == 8:
                print("turbo K=%d snr=%.1f nit=%d bits %s llr %s (bad cb %d, llr mism %d) ber=%.3f" % (K, snr, nit, okb, okl,
                      int(np.any(ref != out, axis=1).sum()), int((ref_llr != out_llr).sum()), np.mean(np.unpackbits(ref, axis=1) != msgs)), flush=True)
        dec.free()
# ---- ldpc
for bg, Z in [(0, 384), (0, 16), (1, 384), (1, 3), (0, 208), (1, 52), (0, 2)]:
    for snr, sf, nit in [(2.0, 0.8, 20), (-1.0, 0.75, 6)]:
        n_cw = 9
        msgs, llrs = O.ldpc_llrs(bg, Z, n_cw, snr, seed=Z + bg, clip=127 if snr < 0 else 63)
        ref, _ = O.ldpc_decode(bg, Z, llrs, sf, nit)
        dec = S.LdpcBatch(bg, Z, sf, nit, n_cw)
        out = dec.decode(llrs)
        ok = np.array_equal(ref, out); ok_all &= ok
        print("ldpc BG%d Z=%d snr=%.1f sf=%.2f it=%d %s (bad cw %d) msgerr=%d" % (bg + 1, Z, snr, sf, nit, ok, int(np.any(ref != out, axis=1).sum()), int((ref != msgs).sum())), flush=True)
        dec.free()
# ---- ofdm
rng = np.random.default_rng(5)
for prb, N, cp, norm, fs, wo, kd in [(6, 0, 0, True, 0, 0, False), (100, 2048, 0, True, 0, 0, False), (100, 0, 0, False, -0.5, 0.5, False), (25, 0, 1, True, 0, 0, False),
                                     (273, 4096, 0, True, 0, 0, True), (50, 0, 0, True, 0.5, 0, False), (15, 0, 0, True, 0, 0, False), (25, 512, 0, True, 0, 0, False), (75, 0, 0, True, 0, 0, False), (200, 3072, 0, True, 0, 0, True)]:
    cfg = O.ofdm_cfg(prb, N, cp, int(norm), fs, wo, int(kd))
    n, nsym, sf_sz, sf_re = O.ofdm_geometry(cfg)
    n_sf = 3
    re = (rng.uniform(-1, 1, (n_sf, sf_re)) + 1j * rng.uniform(-1, 1, (n_sf, sf_re))).astype(np.complex64)
    t_ref = O.ofdm_tx(cfg, re)
    tx = S.OfdmBatch(prb, True, N, cp, norm, fs, wo, kd)
    t_gpu = tx.process(re)
    sc = max(1.0, float(np.sqrt(np.mean(np.abs(t_ref) ** 2))))
    e_tx = np.abs(t_gpu - t_ref).max() / sc
    x = (rng.standard_normal((n_sf, sf_sz)) + 1j * rng.standard_normal((n_sf, sf_sz))).astype(np.complex64) * 0.7
    r_ref = O.ofdm_rx(cfg, x)
    rx = S.OfdmBatch(prb, False, N, cp, norm, fs, wo, kd)
    r_gpu = rx.process(x)
    sc = max(1.0, float(np.sqrt(np.mean(np.abs(r_ref) ** 2))))
    e_rx = np.abs(r_gpu - r_ref).max() / sc
    ok = e_tx < 1e-4 and e_rx < 1e-4; ok_all &= ok
    print("ofdm prb=%d N=%d cp=%d norm=%d fs=%.1f wo=%.1f kd=%d  tx err %.2e  rx err %.2e %s" % (prb, n, cp, norm, fs, wo, kd, e_tx, e_rx, ok), flush=True)
print("ALL OK" if ok_all else "SOME FAILED")
