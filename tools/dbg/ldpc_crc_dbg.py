import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import srslte_amd as S
from srslte_amd import capi
lib = S.lib()
for bg, Z in ((0, 384), (0, 36), (1, 16)):
    N = (68 if bg == 0 else 52) * Z; K = (22 if bg == 0 else 10) * Z
    dec = S.LdpcBatch(bg, Z, 0.8, 8, 4, capi.LDPC_C)
    d_llr = S.DeviceBuffer.from_numpy(np.zeros((4, N), np.int8)); d_msg = S.DeviceBuffer(4 * K); d_it = S.DeviceBuffer(64)
    r = lib.srsran_hip_ldpc_batch_run_crc(dec._h, d_llr.ptr, N - 2 * Z, d_msg.ptr, K, 4, N - 2 * Z, 0x1800063, 24, d_it.ptr, None)
    print(bg, Z, "rc", r, capi.last_error())
