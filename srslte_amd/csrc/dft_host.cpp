// dft_host.cpp -- host side of the generic DFT: srsran_dft_* plan ABI, batch object, transform precoding.
//
// Mirrors (interface + behaviour) lib/src/phy/dft/dft_fftw.c:79-404 and lib/src/phy/dft/dft_precoding.c.
// There is no planner and no wisdom file: a "plan" is a radix list + an exact twiddle table in HBM.
#include "dft_device.h"
#include "hip_common.h"

#include <cmath>
#include <complex>
#include <vector>

using namespace phyhip;

#define DFT_MAX_POINTS 4096

namespace {

struct DftCtx {
  DeviceTag tag;
  int         N = 0;
  int         npass = 0;
  int         radix[16] = {};
  float2*     d_tw = nullptr;
  float2*     d_in = nullptr;
  float2*     d_out = nullptr;
  cf_t*       h_in = nullptr;  // pinned staging
  cf_t*       h_out = nullptr; // pinned staging
  size_t      cap_in = 0, cap_out = 0;
  hipStream_t stream = nullptr;
  // N > 4096: four-step decomposition N = N1 * N2, both <= 4096 (N1 transforms of length N2 after N2 of length N1)
  int     N1 = 0, N2 = 0;
  int     npass1 = 0, npass2 = 0, radix1[16] = {}, radix2[16] = {};
  float2* d_tw1 = nullptr;
  float2* d_tw2 = nullptr;
  float2* d_tmp[2] = {nullptr, nullptr};
  // guru geometry
  cf_t* g_in = nullptr;
  cf_t* g_out = nullptr;
  int   istride = 1, ostride = 1, how_many = 1, idist = 0, odist = 0;
};

void factorize(int N, int* radix, int* npass)
{
  int n = N, np = 0;
  auto take = [&](int r) {
    while (n % r == 0 && np < 16) {
      radix[np++] = r;
      n /= r;
    }
  };
  take(16);
  take(8);
  take(4);
  take(2);
  take(3);
  take(5);
  *npass = (n == 1) ? np : 0; // anything else: direct evaluation
}

int upload_twiddles(float2** d, int N)
{
  std::vector<std::complex<float>> tw(N);
  for (int i = 0; i < N; i++) {
    double a = -2.0 * M_PI * (double)i / (double)N;
    tw[i]    = std::complex<float>((float)cos(a), (float)sin(a));
  }
  (void)hipFree(*d);
  *d = nullptr;
  PHY_HIP_CHECK(hipMalloc(d, N * sizeof(float2)), SRSRAN_ERROR);
  PHY_HIP_CHECK(upload(*d, tw.data(), N * sizeof(float2)), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

bool smooth(int n)
{
  for (int f : {2, 3, 5}) {
    while (n % f == 0) {
      n /= f;
    }
  }
  return n == 1;
}

// N > DFT_MAX_POINTS: split N = N1 * N2 with both factors within the single-kernel range, preferring factors the
// Stockham passes handle (2^a 3^b 5^c) over ones that fall back to direct evaluation
int ctx_set_large(DftCtx* c, int N)
{
  int best = 0;
  for (int pass = 0; pass < 2 && !best; pass++) {
    for (int d = DFT_MAX_POINTS; d >= 2; d--) {
      if (N % d == 0 && N / d <= DFT_MAX_POINTS && (pass == 1 || (smooth(d) && smooth(N / d)))) {
        best = d;
        break;
      }
    }
    if (!best && pass == 0) {
      for (int d = DFT_MAX_POINTS; d >= 2; d--) { // at least one smooth factor
        if (N % d == 0 && N / d <= DFT_MAX_POINTS && smooth(d)) {
          best = d;
          break;
        }
      }
    }
  }
  if (!best) {
    set_error("DFT length %d has no factorisation N1 x N2 with both factors <= %d", N, DFT_MAX_POINTS);
    fprintf(stderr, "[srsran_phy_hip] %s\n", get_error());
    return SRSRAN_ERROR;
  }
  c->N  = N;
  c->N1 = best;
  c->N2 = N / best;
  factorize(c->N1, c->radix1, &c->npass1);
  factorize(c->N2, c->radix2, &c->npass2);
  if (upload_twiddles(&c->d_tw1, c->N1) || upload_twiddles(&c->d_tw2, c->N2)) {
    return SRSRAN_ERROR;
  }
  for (auto& t : c->d_tmp) {
    (void)hipFree(t);
    t = nullptr;
    PHY_HIP_CHECK(hipMalloc(&t, (size_t)N * sizeof(float2)), SRSRAN_ERROR);
  }
  return SRSRAN_SUCCESS;
}

int ctx_set_size(DftCtx* c, int N)
{
  if (N <= 0 || (long)N > (long)DFT_MAX_POINTS * DFT_MAX_POINTS) {
    set_error("DFT length %d is outside the range supported by the HIP engine (1..%d^2)", N, DFT_MAX_POINTS);
    fprintf(stderr, "[srsran_phy_hip] %s\n", get_error());
    return SRSRAN_ERROR;
  }
  if (c->N == N) {
    return SRSRAN_SUCCESS;
  }
  c->N1 = c->N2 = 0;
  if (N > DFT_MAX_POINTS) {
    return ctx_set_large(c, N);
  }
  c->N = N;
  factorize(N, c->radix, &c->npass);
  std::vector<std::complex<float>> tw(N);
  for (int i = 0; i < N; i++) {
    double a = -2.0 * M_PI * (double)i / (double)N;
    tw[i]    = std::complex<float>((float)cos(a), (float)sin(a));
  }
  (void)hipFree(c->d_tw);
  c->d_tw = nullptr;
  PHY_HIP_CHECK(hipMalloc(&c->d_tw, N * sizeof(float2)), SRSRAN_ERROR);
  PHY_HIP_CHECK(upload(c->d_tw, tw.data(), N * sizeof(float2)), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

int ctx_reserve(DftCtx* c, size_t n_in, size_t n_out)
{
  if (n_in > c->cap_in) {
    (void)hipFree(c->d_in);
    (void)hipHostFree(c->h_in);
    PHY_HIP_CHECK(hipMalloc(&c->d_in, n_in * sizeof(float2)), SRSRAN_ERROR);
    PHY_HIP_CHECK(host_image_alloc(&c->h_in, n_in * sizeof(cf_t)), SRSRAN_ERROR);
    c->cap_in = n_in;
  }
  if (n_out > c->cap_out) {
    (void)hipFree(c->d_out);
    (void)hipHostFree(c->h_out);
    PHY_HIP_CHECK(hipMalloc(&c->d_out, n_out * sizeof(float2)), SRSRAN_ERROR);
    PHY_HIP_CHECK(host_image_alloc(&c->h_out, n_out * sizeof(cf_t)), SRSRAN_ERROR);
    c->cap_out = n_out;
  }
  return SRSRAN_SUCCESS;
}

DftCtx* ctx_new(int N)
{
  if (!device_available()) {
    return nullptr;
  }
  auto* c = new DftCtx;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || ctx_set_size(c, N)) {
    delete c;
    return nullptr;
  }
  return c;
}

void ctx_free(DftCtx* c)
{
  if (!c) {
    return;
  }
  (void)hipFree(c->d_tw);
  (void)hipFree(c->d_tw1);
  (void)hipFree(c->d_tw2);
  (void)hipFree(c->d_tmp[0]);
  (void)hipFree(c->d_tmp[1]);
  (void)hipFree(c->d_in);
  (void)hipFree(c->d_out);
  (void)hipHostFree(c->h_in);
  (void)hipHostFree(c->h_out);
  if (c->stream) {
    (void)hipStreamDestroy(c->stream);
  }
  delete c;
}

void fill_params(dft::Params* p, const DftCtx* c, const void* d_in, void* d_out, bool backward, bool mirror, bool dc,
                 bool norm, bool db)
{
  p->in       = d_in;
  p->out      = d_out;
  p->twiddle  = c->d_tw;
  p->idist    = c->N;
  p->odist    = c->N;
  p->istride  = 1;
  p->ostride  = 1;
  p->how_many = 1;
  p->N        = c->N;
  p->npass    = c->npass;
  for (int i = 0; i < 16; i++) {
    p->radix[i] = c->radix[i];
  }
  p->backward = backward;
  p->mirror   = mirror;
  p->dc       = dc;
  p->db       = db;
  p->norm      = norm ? 1.0f / sqrtf((float)c->N) : 0.0f;
  p->real_mode = 0;
}

// N = N1 * N2 > 4096: x[n1 N2 + n2] -> N2 transforms of length N1 (stride N2) -> twiddle w_N^(k1 n2) -> N1 transforms
// of length N2 -> X[k1 + N1 k2].  mirror / dc act on the full length through a separate re-ordering pass.
int run_large(const DftCtx* c, const float2* d_in, float2* d_out, bool backward, bool mirror, bool dc, bool norm, bool db,
              hipStream_t st)
{
  const float2* src = d_in;
  if (mirror && backward) {
    PHY_HIP_CHECK(dft::launch_large_reorder(d_in, c->d_tmp[1], c->N, true, dc, st), SRSRAN_ERROR);
    src = c->d_tmp[1];
  }
  dft::Params p = {};
  p.in       = src;
  p.out      = c->d_tmp[0];
  p.twiddle  = c->d_tw1;
  p.N        = c->N1;
  p.npass    = c->npass1;
  memcpy(p.radix, c->radix1, sizeof(p.radix));
  p.how_many = c->N2;
  p.istride  = c->N2;
  p.ostride  = c->N2;
  p.idist    = 1;
  p.odist    = 1;
  p.backward = backward;
  PHY_HIP_CHECK(dft::launch(p, st), SRSRAN_ERROR);
  PHY_HIP_CHECK(dft::launch_large_twiddle(c->d_tmp[0], c->N1, c->N2, backward, st), SRSRAN_ERROR);
  const bool post = mirror && !backward;
  p.in       = c->d_tmp[0];
  p.out      = post ? c->d_tmp[1] : d_out;
  p.twiddle  = c->d_tw2;
  p.N        = c->N2;
  p.npass    = c->npass2;
  memcpy(p.radix, c->radix2, sizeof(p.radix));
  p.how_many = c->N1;
  p.istride  = 1;
  p.ostride  = c->N1;
  p.idist    = c->N2;
  p.odist    = 1;
  p.norm     = norm ? 1.0f / sqrtf((float)c->N) : 0.0f;
  p.db       = db;
  PHY_HIP_CHECK(dft::launch(p, st), SRSRAN_ERROR);
  if (post) {
    PHY_HIP_CHECK(dft::launch_large_reorder(c->d_tmp[1], d_out, c->N, false, dc, st), SRSRAN_ERROR);
  }
  return SRSRAN_SUCCESS;
}

DftCtx* ctx_raw(srsran_dft_plan_t* plan)
{
  return reinterpret_cast<DftCtx*>(plan->p);
}
// the plan's context -- nullptr (error reported) when it lives on another device than the calling thread's
DftCtx* ctx_of(srsran_dft_plan_t* plan)
{
  DftCtx* c = ctx_raw(plan);
  return (c && !check_device(c->tag, "srsran_dft")) ? nullptr : c;
}

void plan_defaults(srsran_dft_plan_t* plan, int n, srsran_dft_dir_t dir, bool guru)
{
  plan->size      = n;
  plan->init_size = n;
  plan->mode      = SRSRAN_DFT_COMPLEX;
  plan->dir       = dir;
  plan->forward   = dir == SRSRAN_DFT_FORWARD;
  plan->mirror    = false;
  plan->db        = false;
  plan->norm      = false;
  plan->dc        = false;
  plan->is_guru   = guru;
}

} // namespace

// ------------------------------------------------------------------------------------------------ plan ABI

extern "C" int srsran_dft_plan_c(srsran_dft_plan_t* plan, const int dft_points, srsran_dft_dir_t dir)
{
  DftCtx* c = ctx_new(dft_points);
  if (!c) {
    return -1;
  }
  // dft_fftw.c:117-121: the plan owns one input and one output buffer of dft_points samples
  plan->in  = calloc((size_t)dft_points, sizeof(cf_t));
  plan->out = calloc((size_t)dft_points, sizeof(cf_t));
  plan->p   = c;
  plan_defaults(plan, dft_points, dir, false);
  return 0;
}

extern "C" int srsran_dft_plan(srsran_dft_plan_t* plan, const int dft_points, srsran_dft_dir_t dir, srsran_dft_mode_t mode)
{
  memset(plan, 0, sizeof(srsran_dft_plan_t));
  if (mode == SRSRAN_DFT_COMPLEX) {
    return srsran_dft_plan_c(plan, dft_points, dir);
  }
  return srsran_dft_plan_r(plan, dft_points, dir);
}

// dft_fftw.c:255-277: real <-> half-complex transform (FFTW_R2HC forward, FFTW_HC2R backward)
extern "C" int srsran_dft_plan_r(srsran_dft_plan_t* plan, const int dft_points, srsran_dft_dir_t dir)
{
  DftCtx* c = ctx_new(dft_points);
  if (!c) {
    return -1;
  }
  plan->in  = calloc((size_t)dft_points, sizeof(float));
  plan->out = calloc((size_t)dft_points, sizeof(float));
  plan->p   = c;
  plan_defaults(plan, dft_points, dir, false);
  plan->mode = SRSRAN_REAL;
  return 0;
}

extern "C" int srsran_dft_replan_r(srsran_dft_plan_t* plan, const int new_dft_points)
{
  return srsran_dft_replan_c(plan, new_dft_points); // same engine; the transform kind is a run-time flag
}

extern "C" void srsran_dft_run_r(srsran_dft_plan_t* plan, const float* in, float* out)
{
  DftCtx* c = ctx_of(plan);
  if (!c || c->N1 || ctx_reserve(c, plan->size, plan->size)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_dft_run_r: plan not initialised (real transforms: up to %d points)\n", DFT_MAX_POINTS);
    return;
  }
  const size_t bytes = (size_t)plan->size * sizeof(float);
  memcpy(c->h_in, in, bytes);
  // (one transform per call: the kernel loads its input once and stores its output once, so it works on the pinned host images themselves --
  // no copy operation on either side, 6-9 us each whatever the size: tools/probe/roundtrip_probe.hip)
  dft::Params p;
  fill_params(&p, c, c->h_in, c->h_out, !plan->forward, false, false, false, plan->db);
  p.real_mode = plan->forward ? 1 : 2;
  p.norm      = plan->norm ? 1.0f / (float)plan->size : 0.0f; // dft_fftw.c:374-377: 1/N, not 1/sqrt(N)
  PHY_HIP_CHECK_VOID(dft::launch(p, c->stream));
  PHY_HIP_CHECK_VOID(hipStreamSynchronize(c->stream));
  memcpy(out, c->h_out, bytes);
}

extern "C" int srsran_dft_replan_c(srsran_dft_plan_t* plan, const int new_dft_points)
{
  if (plan->size == new_dft_points) {
    return 0;
  }
  DftCtx* c = ctx_of(plan);
  if (!c || ctx_set_size(c, new_dft_points)) {
    return -1;
  }
  plan->size = new_dft_points;
  return 0;
}

extern "C" int srsran_dft_replan(srsran_dft_plan_t* plan, const int new_dft_points)
{
  if (new_dft_points <= plan->init_size) {
    if (plan->mode == SRSRAN_DFT_COMPLEX) {
      return srsran_dft_replan_c(plan, new_dft_points);
    }
    return srsran_dft_replan_r(plan, new_dft_points);
  }
  fprintf(stderr, "DFT: Error calling replan: new_dft_points (%d) must be lower or equal dft_size passed initially (%d)\n",
          new_dft_points, plan->init_size);
  return -1;
}

extern "C" int srsran_dft_plan_guru_c(srsran_dft_plan_t* plan, const int dft_points, srsran_dft_dir_t dir, cf_t* in_buffer,
                                      cf_t* out_buffer, int istride, int ostride, int how_many, int idist, int odist)
{
  DftCtx* c = ctx_new(dft_points);
  if (!c) {
    return -1;
  }
  c->g_in     = in_buffer;
  c->g_out    = out_buffer;
  c->istride  = istride;
  c->ostride  = ostride;
  c->how_many = how_many;
  c->idist    = idist;
  c->odist    = odist;
  plan->p     = c;
  plan_defaults(plan, dft_points, dir, true);
  return 0;
}

extern "C" int srsran_dft_replan_guru_c(srsran_dft_plan_t* plan, const int new_dft_points, cf_t* in_buffer, cf_t* out_buffer,
                                        int istride, int ostride, int how_many, int idist, int odist)
{
  DftCtx* c = ctx_of(plan);
  if (!c || ctx_set_size(c, new_dft_points)) {
    return -1;
  }
  c->g_in         = in_buffer;
  c->g_out        = out_buffer;
  c->istride      = istride;
  c->ostride      = ostride;
  c->how_many     = how_many;
  c->idist        = idist;
  c->odist        = odist;
  plan->size      = new_dft_points;
  plan->init_size = plan->size; // dft_fftw.c:144-145
  return 0;
}

extern "C" void srsran_dft_plan_free(srsran_dft_plan_t* plan)
{
  if (!plan) {
    return;
  }
  if (!plan->size) {
    return; // dft_fftw.c:389-391
  }
  if (!plan->is_guru) {
    free(plan->in);
    free(plan->out);
  }
  ctx_free(ctx_raw(plan));
  memset(plan, 0, sizeof(srsran_dft_plan_t));
}

extern "C" void srsran_dft_plan_set_mirror(srsran_dft_plan_t* plan, bool val)
{
  plan->mirror = val;
}
extern "C" void srsran_dft_plan_set_db(srsran_dft_plan_t* plan, bool val)
{
  plan->db = val;
}
extern "C" void srsran_dft_plan_set_norm(srsran_dft_plan_t* plan, bool val)
{
  plan->norm = val;
}
extern "C" void srsran_dft_plan_set_dc(srsran_dft_plan_t* plan, bool val)
{
  plan->dc = val;
}

static void run_contiguous(srsran_dft_plan_t* plan, const cf_t* in, cf_t* out, bool options)
{
  DftCtx* c = ctx_of(plan);
  if (!c || ctx_reserve(c, plan->size, plan->size)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_dft_run: plan not initialised\n");
    return;
  }
  const size_t bytes = (size_t)plan->size * sizeof(cf_t);
  memcpy(c->h_in, in, bytes);
  // copy_post leaves the last `dc` output samples untouched: seed the staging buffer with the caller's data
  memcpy(c->h_out, out, bytes);
  if (c->N1) {
    // four-step path: several passes over the data, which therefore lives on the device
    PHY_HIP_CHECK_VOID(hipMemcpyAsync(c->d_in, c->h_in, bytes, hipMemcpyHostToDevice, c->stream));
    PHY_HIP_CHECK_VOID(hipMemcpyAsync(c->d_out, c->h_out, bytes, hipMemcpyHostToDevice, c->stream));
    if (run_large(c, c->d_in, c->d_out, !plan->forward, options && plan->mirror, options && plan->dc, options && plan->norm,
                  options && plan->db, c->stream)) {
      fprintf(stderr, "[srsran_phy_hip] srsran_dft_run: %s\n", get_error());
      return;
    }
    PHY_HIP_CHECK_VOID(hipMemcpyAsync(c->h_out, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));
  } else {
    // one kernel, input loaded once, output stored once: on the pinned host images themselves (see srsran_dft_run_r)
    dft::Params p;
    fill_params(&p, c, c->h_in, c->h_out, !plan->forward, options && plan->mirror, options && plan->dc, options && plan->norm,
                options && plan->db);
    PHY_HIP_CHECK_VOID(dft::launch(p, c->stream));
  }
  PHY_HIP_CHECK_VOID(hipStreamSynchronize(c->stream));
  memcpy(out, c->h_out, bytes);
}

extern "C" void srsran_dft_run_c(srsran_dft_plan_t* plan, const cf_t* in, cf_t* out)
{
  run_contiguous(plan, in, out, true);
}

extern "C" void srsran_dft_run_c_zerocopy(srsran_dft_plan_t* plan, const cf_t* in, cf_t* out)
{
  run_contiguous(plan, in, out, false); // fftwf_execute_dft: no mirror/dc/norm (dft_fftw.c:331-334)
}

extern "C" void srsran_dft_run(srsran_dft_plan_t* plan, const void* in, void* out)
{
  if (plan->mode == SRSRAN_DFT_COMPLEX) {
    srsran_dft_run_c(plan, (const cf_t*)in, (cf_t*)out);
  } else {
    srsran_dft_run_r(plan, (const float*)in, (float*)out);
  }
}

extern "C" void srsran_dft_run_guru_c(srsran_dft_plan_t* plan)
{
  if (!plan->is_guru) {
    fprintf(stderr, "srsran_dft_run_guru_c: the selected plan is not guru!\n");
    return;
  }
  DftCtx* c = ctx_of(plan);
  if (!c || !c->g_in || !c->g_out) {
    return;
  }
  if (c->N1) {
    // one contiguous transform longer than 4096 points goes through the four-step path
    if (c->how_many != 1 || c->istride != 1 || c->ostride != 1) {
      fprintf(stderr, "[srsran_phy_hip] srsran_dft_run_guru_c: strided / batched plans are limited to %d points\n", DFT_MAX_POINTS);
      return;
    }
    const size_t bytes = (size_t)plan->size * sizeof(cf_t);
    if (ctx_reserve(c, plan->size, plan->size)) {
      return;
    }
    memcpy(c->h_in, c->g_in, bytes);
    PHY_HIP_CHECK_VOID(hipMemcpyAsync(c->d_in, c->h_in, bytes, hipMemcpyHostToDevice, c->stream));
    if (run_large(c, c->d_in, c->d_out, !plan->forward, false, false, false, false, c->stream)) {
      return;
    }
    PHY_HIP_CHECK_VOID(hipMemcpyAsync(c->h_out, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    PHY_HIP_CHECK_VOID(hipStreamSynchronize(c->stream));
    memcpy(c->g_out, c->h_out, bytes);
    return;
  }
  const int    N       = plan->size;
  const size_t span_in  = (size_t)(c->how_many - 1) * c->idist + (size_t)(N - 1) * c->istride + 1;
  const size_t span_out = (size_t)(c->how_many - 1) * c->odist + (size_t)(N - 1) * c->ostride + 1;
  if (ctx_reserve(c, span_in, span_out)) {
    return;
  }
  memcpy(c->h_in, c->g_in, span_in * sizeof(cf_t));
  PHY_HIP_CHECK_VOID(hipMemcpyAsync(c->d_in, c->h_in, span_in * sizeof(cf_t), hipMemcpyHostToDevice, c->stream));
  dft::Params p;
  fill_params(&p, c, c->d_in, c->d_out, !plan->forward, false, false, false, false);
  p.idist    = c->idist;
  p.odist    = c->odist;
  p.istride  = c->istride;
  p.ostride  = c->ostride;
  p.how_many = c->how_many;
  PHY_HIP_CHECK_VOID(dft::launch(p, c->stream));
  PHY_HIP_CHECK_VOID(hipMemcpyAsync(c->h_out, c->d_out, span_out * sizeof(cf_t), hipMemcpyDeviceToHost, c->stream));
  PHY_HIP_CHECK_VOID(hipStreamSynchronize(c->stream));
  // write only the samples the transform produces (gaps between transforms belong to the caller)
  for (int b = 0; b < c->how_many; b++) {
    if (c->ostride == 1) {
      memcpy(c->g_out + (size_t)b * c->odist, c->h_out + (size_t)b * c->odist, (size_t)N * sizeof(cf_t));
    } else {
      for (int k = 0; k < N; k++) {
        c->g_out[(size_t)b * c->odist + (size_t)k * c->ostride] = c->h_out[(size_t)b * c->odist + (size_t)k * c->ostride];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ batch object

struct srsran_hip_dft_batch {
  DftCtx* c = nullptr;
  bool    backward = false, mirror = false, dc = false, norm = false;
};

extern "C" int srsran_hip_dft_batch_create(srsran_hip_dft_batch_t** hh, int dft_points, srsran_dft_dir_t dir, bool mirror,
                                           bool dc, bool norm)
{
  if (!hh) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  *hh       = nullptr;
  if (dft_points > DFT_MAX_POINTS) {
    set_error("dft batch: batched transforms are limited to %d points", DFT_MAX_POINTS);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  DftCtx* c = ctx_new(dft_points);
  if (!c) {
    return SRSRAN_ERROR;
  }
  auto* h     = new srsran_hip_dft_batch;
  h->c        = c;
  h->backward = dir == SRSRAN_DFT_BACKWARD;
  h->mirror   = mirror;
  h->dc       = dc;
  h->norm     = norm;
  *hh         = h;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_hip_dft_batch_free(srsran_hip_dft_batch_t* h)
{
  if (h) {
    ctx_free(h->c);
    delete h;
  }
}

extern "C" int srsran_hip_dft_batch_run(srsran_hip_dft_batch_t* h, const cf_t* d_in, cf_t* d_out, uint32_t how_many, void* stream)
{
  if (!h || !d_in || !d_out || how_many == 0) {
    set_error("dft batch: invalid arguments");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  PHY_DEV_GUARD(h->c->tag, "srsran_hip_dft_batch_run", SRSRAN_ERROR);
  dft::Params p;
  fill_params(&p, h->c, d_in, d_out, h->backward, h->mirror, h->dc, h->norm, false);
  p.how_many = (int)how_many;
  PHY_HIP_CHECK(dft::launch(p, (hipStream_t)stream), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ transform precoding

extern "C" bool srsran_dft_precoding_valid_prb(uint32_t nof_prb)
{
  // dft_precoding.c:84-96: 2^a 3^b 5^c up to 100 PRB (TS 36.211 5.3.3); the table's entry 0 is `true`
  if (nof_prb > 100) {
    return false;
  }
  uint32_t n = nof_prb;
  if (n == 0) {
    return true;
  }
  while (n % 2 == 0) {
    n /= 2;
  }
  while (n % 3 == 0) {
    n /= 3;
  }
  while (n % 5 == 0) {
    n /= 5;
  }
  return n == 1;
}

extern "C" uint32_t srsran_dft_precoding_get_valid_prb(uint32_t nof_prb)
{
  while (!srsran_dft_precoding_valid_prb(nof_prb)) {
    nof_prb--;
  }
  return nof_prb;
}

extern "C" void srsran_dft_precoding_free(srsran_dft_precoding_t* q)
{
  for (uint32_t i = 1; i <= q->max_prb; i++) {
    if (srsran_dft_precoding_valid_prb(i)) {
      srsran_dft_plan_free(&q->dft_plan[i]);
    }
  }
  memset(q, 0, sizeof(srsran_dft_precoding_t));
}

extern "C" int srsran_dft_precoding_init(srsran_dft_precoding_t* q, uint32_t max_prb, bool is_tx)
{
  int ret = SRSRAN_ERROR_INVALID_INPUTS;
  memset(q, 0, sizeof(srsran_dft_precoding_t));
  if (max_prb <= SRSRAN_MAX_PRB) {
    ret = SRSRAN_ERROR;
    for (uint32_t i = 1; i <= max_prb; i++) {
      if (srsran_dft_precoding_valid_prb(i)) {
        if (srsran_dft_plan_c(&q->dft_plan[i], (int)(i * 12), is_tx ? SRSRAN_DFT_FORWARD : SRSRAN_DFT_BACKWARD)) {
          fprintf(stderr, "Error: Creating DFT plan %d\n", i);
          q->max_prb = i; // so that free() releases what was created
          srsran_dft_precoding_free(q);
          return ret;
        }
        srsran_dft_plan_set_norm(&q->dft_plan[i], true);
      }
    }
    q->max_prb = max_prb;
    ret        = SRSRAN_SUCCESS;
  }
  return ret;
}

extern "C" int srsran_dft_precoding_init_rx(srsran_dft_precoding_t* q, uint32_t max_prb)
{
  return srsran_dft_precoding_init(q, max_prb, false);
}

extern "C" int srsran_dft_precoding_init_tx(srsran_dft_precoding_t* q, uint32_t max_prb)
{
  return srsran_dft_precoding_init(q, max_prb, true);
}

extern "C" int srsran_dft_precoding(srsran_dft_precoding_t* q, cf_t* input, cf_t* output, uint32_t nof_prb, uint32_t nof_symbols)
{
  // NB the reference's guard is `!valid && nof_prb <= max_prb` (dft_precoding.c:118); an invalid size
  // above max_prb would run an unplanned plan there.  We refuse both.
  if (!srsran_dft_precoding_valid_prb(nof_prb) || nof_prb > q->max_prb || nof_prb == 0) {
    fprintf(stderr, "Error invalid number of PRB (%d)\n", nof_prb);
    return SRSRAN_ERROR;
  }
  srsran_dft_plan_t* plan = &q->dft_plan[nof_prb];
  DftCtx*            c    = ctx_of(plan);
  const size_t       n    = (size_t)nof_symbols * 12 * nof_prb;
  if (!c || ctx_reserve(c, n, n)) {
    return SRSRAN_ERROR;
  }
  // all symbols in one launch (the reference loops srsran_dft_run_c per symbol, :122-124)
  memcpy(c->h_in, input, n * sizeof(cf_t));
  dft::Params p; // (the kernel works on the pinned host images themselves, see srsran_dft_run_r)
  fill_params(&p, c, c->h_in, c->h_out, !plan->forward, plan->mirror, plan->dc, plan->norm, plan->db);
  p.how_many = (int)nof_symbols;
  PHY_HIP_CHECK(dft::launch(p, c->stream), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(c->stream), SRSRAN_ERROR);
  memcpy(output, c->h_out, n * sizeof(cf_t));
  return SRSRAN_SUCCESS;
}
