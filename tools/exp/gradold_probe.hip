// gradold_probe.hip -- probe library only (`make probe`): the gradient reduction EXACTLY as it was before round 2 removed its
// 32- and 64-accumulator instances (tools/exp/gradold/ = `git show e342069^:gogp_amd/csrc/{grad.hip,common.h,
// kern_eval.h}`, verbatim), compiled into its own namespace so that the faulty instance
// grad_reduce_kernel<64, true, double> can be run against today's kernels (gogp_test_grad64, variant 1).
#define gogp gogp_old
#include "gradold/grad.hip"
#undef gogp

extern "C" void gogp_old_grad_reduce_local(hipStream_t s, const void *devparams, int ndim, int ard_dims, const double *X,
                                           const double *alpha, const double *Kinv, long ld, long n, long mrows,
                                           long ncols, double *partials, double *out) {
  gogp_old::BlockMap map;  // 1 x 1 grid
  gogp_old::launch_grad_reduce_local(s, (const gogp_old::DevParams *)devparams, ndim, ard_dims, X, alpha, Kinv, ld, n, mrows,
                                     ncols, map, partials, out);
}
// the faulty instance with a chosen number of workgroups (<= 256: every workgroup is the first to use the registers
// of its SIMDs after a scrub)
extern "C" void gogp_old_grad64_local_blocks(hipStream_t s, const void *devparams, int ndim, const double *X,
                                             const double *alpha, const double *Kinv, long ld, long n, long mrows,
                                             long ncols, double *partials, double *out, int blocks) {
  gogp_old::BlockMap map;
  const int nt = (int)(mrows / 64), ntc = (int)(ncols / 64);
  const size_t lds = (size_t)(128 * ndim + 128 + 4 * gogp_old::NACC) * sizeof(double);
  hipLaunchKernelGGL((gogp_old::grad_reduce_kernel<64, true, double>), dim3(blocks), dim3(256), lds, s,
                     (const gogp_old::DevParams *)devparams, X, alpha, Kinv, ld, n, nt, nt * ntc, partials, ntc, map);
  hipLaunchKernelGGL(gogp_old::grad_final_kernel, dim3(gogp_old::NACC), dim3(256), 0, s, partials, blocks, out);
}
extern "C" void gogp_old_grad_reduce(hipStream_t s, const void *devparams, int ndim, int ard_dims, const double *X,
                                     const double *alpha, const double *Kinv, long ld, long n, long npad, double *partials,
                                     double *out) {
  gogp_old::launch_grad_reduce(s, (const gogp_old::DevParams *)devparams, ndim, ard_dims, X, alpha, Kinv, ld, n, npad,
                               partials, out);
}
