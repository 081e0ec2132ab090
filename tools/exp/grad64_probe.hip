// grad64_probe.hip -- forensic probe, built only by `make probe` into libgogp_probe.so (tools/exp/gogp_probe.h: gogp_test_grad64).
//
// Rounds 1-2 shipped 32- and 64-accumulator instances of the per-pair gradient reduction; the 64-accumulator
// LOCAL one returned wrong, run-to-run varying sums as soon as a workgroup walked a second tile, and round 2
// removed every instance above 256 VGPRs.  The kernel template is unchanged, so the
// instance can be rebuilt here -- outside the product library -- to test what the ISA suggests: that it reads
// register lanes it never wrote (AGPR spill copies under a partial EXEC mask).  The experiment: run it as it
// is, then again after a scrub kernel has filled every VGPR and AGPR of every SIMD with a known pattern.  If
// the sums follow the pattern, the kernel consumes stale register contents.
#define GOGP_GRAD_KERNEL_ONLY
#include "../../gogp_amd/csrc/grad.hip"

#include <math.h>
#include <string.h>

#include <vector>

#include "gogp_probe.h"

namespace gogp {

#include "scrub_regs.inc"

// value: the pattern for the AGPRs; the architectural VGPRs get vvalue
static unsigned g_vvalue_same = 1;  // 1: VGPRs get the same pattern as the AGPRs; 0: VGPRs always get 0
static void scrub(hipStream_t s, unsigned value, unsigned *sink) {
  // 512 registers per lane: one wave per SIMD, one workgroup per CU at a time; 4096 workgroups pass over
  // every CU many times
  hipLaunchKernelGGL(scrub_regs_kernel, dim3(4096), dim3(256), 0, s, g_vvalue_same ? value : 0u, value, sink);
}

}  // namespace gogp

using namespace gogp;

// gradold_th.hip: the pre-round-2 source, verbatim, in its own namespace
extern "C" void gogp_old_grad_reduce_local(hipStream_t s, const void *devparams, int ndim, int ard_dims, const double *X,
                                           const double *alpha, const double *Kinv, long ld, long n, long mrows,
                                           long ncols, double *partials, double *out);
extern "C" void gogp_old_grad_reduce(hipStream_t s, const void *devparams, int ndim, int ard_dims, const double *X,
                                     const double *alpha, const double *Kinv, long ld, long n, long npad, double *partials,
                                     double *out);
extern "C" void gogp_old_grad64_local_blocks(hipStream_t s, const void *devparams, int ndim, const double *X,
                                             const double *alpha, const double *Kinv, long ld, long n, long mrows,
                                             long ncols, double *partials, double *out, int blocks);

// out: per run NACC slot sums (fixed-order host sum of the workgroup partials); runs: [0] reference (the
// kept 16-accumulator instances, three passes), [1 .. reps] the 64-accumulator LOCAL instance as it is,
// then per scrub value two runs of it right behind a scrub with that value.
// variant 0: the 64-accumulator LOCAL instance rebuilt from today's kernel template; 1: the pre-round-2 source
// (launcher and kernel, LOCAL form: the one that failed); 2: its unsharded form
extern "C" int gogp_test_grad64(int device, int64_t n, int D, int reps, const unsigned *scrub_values, int nscrub,
                                int variant, double *out) {
  if (n < 64 || D < 33 || D > 64 || reps < 0 || nscrub < 0 || !out) return GOGP_EARG;
  // variant + 10000: the scrub patterns go to the AGPRs only (VGPRs are zeroed); + 20000: to the VGPRs only
  gogp::g_vvalue_same = 1;
  int agpr_only = 0, vgpr_only = 0;
  if (variant >= 20000) {
    variant -= 20000;
    vgpr_only = 1;
  } else if (variant >= 10000) {
    variant -= 10000;
    agpr_only = 1;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GOGP_EHIP;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return GOGP_EHIP;
  const int64_t npad = (n + 511) / 512 * 512;
  std::vector<double> hX((size_t)npad * D + 64, 0.0), ha((size_t)npad, 0.0), hK((size_t)npad * npad, 0.0);
  unsigned long long st = 88172645463325252ULL;
  auto rnd = [&]() {
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    return (double)(st >> 11) / 9007199254740992.0;
  };
  for (int64_t i = 0; i < n; ++i) {
    for (int d = 0; d < D; ++d) hX[(size_t)i * D + d] = rnd();
    ha[(size_t)i] = rnd() - 0.5;
  }
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j <= i; ++j) hK[(size_t)i * npad + j] = (rnd() - 0.5) * (i == j ? 4.0 : 0.05);
  DevParams hp;
  memset(&hp, 0, sizeof hp);
  hp.ndim = D;
  hp.nterms = 1;
  hp.ns = D + 1;
  hp.nn = 1;
  hp.kind[0] = GOGP_K_NORMAL;
  hp.ard[0] = 1;
  hp.c[0] = 1.1;
  for (int d = 0; d < D; ++d) hp.inv_len[0][d] = 1.0 / (2.5 + 0.03 * d);
  hp.noise_var = 0.04;
  hp.dnoise = 0.08;
  double *dX = nullptr, *da = nullptr, *dK = nullptr, *dpart = nullptr, *dout = nullptr;
  DevParams *dp = nullptr;
  unsigned *dsink = nullptr;
  const int nt = (int)(npad / 64);
  const int blocks = grad_reduce_blocks_local(npad, npad);
  hipError_t e = hipMalloc(&dX, hX.size() * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&da, ha.size() * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&dK, hK.size() * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&dpart, (size_t)blocks * NACC * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&dout, NACC * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&dp, sizeof(DevParams));
  if (e == hipSuccess) e = hipMalloc(&dsink, 64);
  if (e == hipSuccess) e = hipMemcpy(dX, hX.data(), hX.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(da, ha.data(), ha.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dK, hK.data(), hK.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dp, &hp, sizeof hp, hipMemcpyHostToDevice);
  int rc = (e == hipSuccess) ? GOGP_OK : GOGP_ENOMEM;
  BlockMap map;  // 1 x 1 grid: local == global
  const size_t lds = (size_t)(128 * D + 64 * 64 + 128 + 4 * NACC) * sizeof(double);  // D + ARD_D (= 64) rows of CjT
  std::vector<double> hpart((size_t)blocks * NACC);
  int run = 0;
  auto collect = [&]() {  // fixed-order host sum of the partials of the 64-accumulator launch
    if (variant != 0) {  // the old launchers end with their own final reduction
      if (hipMemcpy(out + (size_t)run * NACC, dout, NACC * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = GOGP_EHIP;
      ++run;
      return;
    }
    if (hipMemcpy(hpart.data(), dpart, hpart.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = GOGP_EHIP;
    for (int q = 0; q < NACC; ++q) {
      double s = 0.0;
      for (int b = 0; b < blocks; ++b) s += hpart[(size_t)b * NACC + q];
      out[(size_t)run * NACC + q] = s;
    }
    ++run;
  };
  auto launch64 = [&]() {
    if (variant >= 16)  // the pre-round-2 sharded instance on `variant` workgroups
      gogp_old_grad64_local_blocks(0, dp, D, dX, da, dK, (long)npad, (long)n, (long)npad, (long)npad, dpart, dout, variant);
    else if (variant == 1)
      gogp_old_grad_reduce_local(0, dp, D, D, dX, da, dK, (long)npad, (long)n, (long)npad, (long)npad, dpart, dout);
    else if (variant == 2)
      gogp_old_grad_reduce(0, dp, D, D, dX, da, dK, (long)npad, (long)n, (long)npad, dpart, dout);
    else
      hipLaunchKernelGGL((grad_reduce_kernel<64, true, double, false>), dim3(blocks), dim3(256), lds, 0, dp, dX, da, dK,
                         (long)npad, (long)n, nt, nt * nt, dpart, nt, map, 0, 0L);
  };
  if (rc == GOGP_OK) {
    // reference: the product library's kept instances (per-pair kernel, passes of 16 dimensions)
    launch_grad_reduce_local(0, dp, D, D, dX, da, dK, npad, n, npad, npad, map, dpart, dout, false, 65);
    if (hipMemcpy(out, dout, NACC * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = GOGP_EHIP;
    ++run;
    for (int r = 0; r < reps && rc == GOGP_OK; ++r) {
      launch64();
      collect();
    }
    for (int k = 0; k < nscrub && rc == GOGP_OK; ++k)
      for (int twice = 0; twice < 2; ++twice) {
        if (agpr_only)
          hipLaunchKernelGGL(scrub_regs_kernel, dim3(4096), dim3(256), 0, 0, 0u, scrub_values[k], dsink);
        else if (vgpr_only)
          hipLaunchKernelGGL(scrub_regs_kernel, dim3(4096), dim3(256), 0, 0, scrub_values[k], 0u, dsink);
        else
          scrub(0, scrub_values[k], dsink);
        launch64();
        collect();
      }
  }
  if (hipDeviceSynchronize() != hipSuccess) rc = GOGP_EHIP;
  for (void *p : {(void *)dX, (void *)da, (void *)dK, (void *)dpart, (void *)dout, (void *)dp, (void *)dsink}) (void)hipFree(p);
  (void)hipGetLastError();
  return rc;
}
