import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
lib = S.lib()
h = C.c_void_p(); capi.check(lib.srsran_hip_sch_enc_create(C.byref(h)), "c")
rng = np.random.default_rng(0)
for tbs, Qm, G in ((40, 2, 120), (6200, 6, 9000)):
    payload = rng.integers(0, 256, tbs // 8).astype(np.uint8)
    d_data = S.DeviceBuffer.from_numpy(payload); d_e = S.DeviceBuffer.from_numpy(np.zeros(G // 8 + 8, np.uint8))
    tb = (capi.HipTb * 1)(capi.HipTb(tbs, Qm, 0, G, 0, 0, 0))
    capi.check(lib.srsran_hip_sch_encode(h, d_data.ptr, tb, 1, d_e.ptr, None), "e"); capi.check(lib.srsran_hip_stream_sync(None), "s")
    got = np.unpackbits(d_e.to_numpy(np.uint8, (G // 8 + 8,)))[:G]
    want, _ = O.tb_coded_bits(tbs, Qm, G, 0, None, payload=np.unpackbits(payload), tx_order=True)
    bad = np.nonzero(got != want)[0]
    K = O.cbsegm(tbs)["K1"]
    t = O.rm_table(K, 0)
    src = t[bad % t.size]
    print(tbs, "mismatches", bad.size, "of", G, "src stream/idx:", [(int(s) % 3, int(s) // 3) if s < 3 * K else ("tail", int(s) - 3 * K) for s in src[:20]])
# reconstruct the device's d from a transmission that covers the whole circular buffer
tbs, Qm = 40, 2
G = 3 * (64 + 4) * 2
payload = rng.integers(0, 256, tbs // 8).astype(np.uint8)
d_data = S.DeviceBuffer.from_numpy(payload); d_e = S.DeviceBuffer.from_numpy(np.zeros(G // 8 + 8, np.uint8))
tb = (capi.HipTb * 1)(capi.HipTb(tbs, Qm, 0, G, 0, 0, 0))
capi.check(lib.srsran_hip_sch_encode(h, d_data.ptr, tb, 1, d_e.ptr, None), "e"); capi.check(lib.srsran_hip_stream_sync(None), "s")
got = np.unpackbits(d_e.to_numpy(np.uint8, (G // 8 + 8,)))[:G]
K = 64
t = O.rm_table(K, 0)
d_dev = np.zeros(3 * K + 12, np.uint8); d_dev[t] = got[:t.size]
cb = O.crc_attach(np.unpackbits(payload), O.CRC24A)
d_orc = O.turbo_encode(cb)
print("c equal", np.array_equal(d_dev[0:3*K:3], d_orc[0:3*K:3]), "p1 equal", np.array_equal(d_dev[1:3*K:3], d_orc[1:3*K:3]), "p2 equal", np.array_equal(d_dev[2:3*K:3], d_orc[2:3*K:3]))
print("p2 dev", d_dev[2:3*K:3]); print("p2 orc", d_orc[2:3*K:3]); print("tail", d_dev[3*K:], d_orc[3*K:])
