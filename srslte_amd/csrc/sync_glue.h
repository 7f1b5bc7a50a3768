// sync_glue.h -- host-pointer helpers shared by the srsran_sync_t glue and the PSS extras: every one of them
// stages its operands to the device, runs a kernel of sync_kernels.hip and copies the result back.  They
// stand where the reference calls its srsran_vec_* SIMD routines on a single frame; none is a throughput path.
#pragma once
#include "hip_common.h"

namespace phyhip {
namespace glue {

// out[i] = a[i] * b[i]  (or a[i] * conj(b[i]))            srsran_vec_prod_ccc / srsran_vec_prod_conj_ccc
int prod(const cf_t* a, const cf_t* b, cf_t* out, int n, bool conj_b);
// out[i] = sa * a[i] + sb * b[i]
int lincomb(const cf_t* a, float sa, const cf_t* b, float sb, cf_t* out, int n);

enum { PLAIN = 0, CONJ = 1, POWER = 2 }; // sum a*b | sum a*conj(b) | mean |a|^2 (real part of the result)
struct Dot {
  const cf_t* a;
  const cf_t* b; // unused for POWER
  int         n;
  int         mode;
};
// up to 8 reductions in one launch
int dots(const Dot* jobs, int count, cf_t* results);

// srsran_vec_apply_cfo (vector_simd.c:1692-1739): out[i] = in[i] e^{j 2 pi cfo i}, the reference's recursive oscillators
int apply_cfo(const cf_t* in, cf_t* out, int n, float cfo);

// srsran_cp_synch (cp.c:60-79); n_in = number of input samples the kernel may touch
int cp_synch(const cf_t* in, int n_in, cf_t* corr, int max_offset, int nof_symbols, int cp_len, int N, uint32_t* argmax);

} // namespace glue
} // namespace phyhip
