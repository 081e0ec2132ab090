// persist_probe.hip -- VERDICT round 3, item 1(a), as a measurement: the fp64 tile kernel's large shape (128 x 128 tile,
// 8 waves) with a WORKGROUP LOOP over tiles and the next tile's first operand k-step requested before the current tile's
// stores (cross-tile software pipelining), against the product kernel (one workgroup per tile) on plain RECT launches.
// Not part of the product: a stand-alone executable,
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -I gogp_amd/csrc tools/exp/persist_probe.hip -o tools/exp/persist_probe
//   tools/exp/persist_probe            (on the GPU box)
// It includes the product kernel's source for the helpers and the baseline.
#include <cstdio>
#include <vector>

#include "dgemm.hip"

namespace gogp {

template <int NWG_PER_CU>
__global__ __launch_bounds__(512, 4) void dgemm_persist_kernel(GemmArgs g, int ntiles) {
  constexpr int BT = 128, NW = 8;
  constexpr int MT = BT / 32, NTW = BT / 64, WT = BT / 2, WTN = BT / 4;
  constexpr int NQ = BT * 8 / (NW * 64), SROWS = NW * 8;
  __shared__ __attribute__((aligned(16))) double lds[2][2][BT * GEMM_BK];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int srow = tid >> 3, schunk = tid & 7;
  const int gchunk = schunk ^ ((srow >> 1) & 7);
  const long a_step = (long)SROWS * g.lda, b_step = (long)SROWS * g.ldb;
  const int wr = wid >> 2, wc = wid & 3;
  const int frow = lane & 15, fk = lane >> 4, fchunk = fk >> 1, fhalf = fk & 1;
  const int abase = (wr * WT + frow) * GEMM_BK, bbase = (wc * WTN + frow) * GEMM_BK;
  unsigned afrag[4], bfrag[4];
  {
    const unsigned lds0 = lds_byte_address(&lds[0][0][0]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int xk = (((kk * 2 + fchunk) ^ ((frow >> 1) & 7)) << 1) + fhalf;
      afrag[kk] = lds0 + 8u * (unsigned)(abase + xk);
      bfrag[kk] = lds0 + 8u * (unsigned)(BT * GEMM_BK + bbase + xk);
    }
  }
  constexpr unsigned BUF_BYTES = 2u * BT * GEMM_BK * 8u;
  const int coff = (lane >> 4) * (int)g.ldc + (lane & 15);
  const double alpha = g.alpha, sc = g.beta / g.alpha;
  const int nkt = g.nkt;

  // tile list of this workgroup: t, t + G, t + 2G, ... with the product kernel's XCD-aware remap (G is a multiple of 8,
  // so every tile of a workgroup maps into its own XCD's contiguous chunk)
  auto remap = [&](int t) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
  };
  int t_raw = blockIdx.x;
  if (t_raw >= ntiles) return;
  int t = remap(t_raw);
  int ti = t / g.nt, tj = t - ti * g.nt;
  const double *Ap = g.A + (long)ti * BT * g.lda + (long)srow * g.lda + gchunk * 2;
  const double *Bp = g.B + (long)tj * BT * g.ldb + (long)srow * g.ldb + gchunk * 2;
  double *Cg = g.C + (long)(ti * BT + wr * WT) * g.ldc + tj * BT + wc * WTN;
  int cur = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    load16_to_lds(Ap + q * a_step, &lds[0][0][(wid * 8 + SROWS * q) * GEMM_BK]);
    load16_to_lds(Bp + q * b_step, &lds[0][1][(wid * 8 + SROWS * q) * GEMM_BK]);
  }
  for (;;) {
    f64x4 acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[m][n][v] = sc * (Cg + (long)(m * 16 + 4 * v) * g.ldc)[coff + n * 16];
    wait_vmcnt0();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const bool more = (kt + 1 < nkt);
      const unsigned curoff = (unsigned)cur * BUF_BYTES;
      double a[2][MT], b[2][NTW];
      read_frags<0, MT>(a[0], afrag[0] + curoff);
      read_frags<0, NTW>(b[0], bfrag[0] + curoff);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) {
          read_frags<0, MT>(a[(kk + 1) & 1], afrag[kk + 1] + curoff);
          read_frags<0, NTW>(b[(kk + 1) & 1], bfrag[kk + 1] + curoff);
          lds_wait<MT + NTW>(a[kk & 1], b[kk & 1]);
        } else {
          lds_wait<0>(a[kk & 1], b[kk & 1]);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NTW; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk & 1][m], b[kk & 1][n], acc[m][n], 0, 0, 0);
        sched_fence();
        if (kk == 0 && more) {
          const double *ap = Ap + (long)(kt + 1) * GEMM_BK;
          const double *bp = Bp + (long)(kt + 1) * GEMM_BK;
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            load16_to_lds(ap + q * a_step, &lds[cur ^ 1][0][(wid * 8 + SROWS * q) * GEMM_BK]);
            load16_to_lds(bp + q * b_step, &lds[cur ^ 1][1][(wid * 8 + SROWS * q) * GEMM_BK]);
          }
          sched_fence();
        }
      }
      if (more) wait_vmcnt0();
      __syncthreads();
      cur ^= 1;
    }
    // ---- tile boundary: the next tile's first k-step goes out BEFORE this tile's stores (both LDS buffers are free
    // behind the last barrier); its C tile is requested right behind the stores
    const int t_next = t_raw + gridDim.x;
    const bool has_next = t_next < ntiles;
    double *Cn = Cg;
    if (has_next) {
      t = remap(t_next);
      ti = t / g.nt;
      tj = t - ti * g.nt;
      Ap = g.A + (long)ti * BT * g.lda + (long)srow * g.lda + gchunk * 2;
      Bp = g.B + (long)tj * BT * g.ldb + (long)srow * g.ldb + gchunk * 2;
      Cn = g.C + (long)(ti * BT + wr * WT) * g.ldc + tj * BT + wc * WTN;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        load16_to_lds(Ap + q * a_step, &lds[cur][0][(wid * 8 + SROWS * q) * GEMM_BK]);
        load16_to_lds(Bp + q * b_step, &lds[cur][1][(wid * 8 + SROWS * q) * GEMM_BK]);
      }
      sched_fence();
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int v = 0; v < 4; ++v) (Cg + (long)(m * 16 + 4 * v) * g.ldc)[coff + n * 16] = alpha * acc[m][n][v];
    if (!has_next) break;
    t_raw = t_next;
    Cg = Cn;
  }
}

}  // namespace gogp

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                          \
      return 1;                                                               \
    }                                                                         \
  } while (0)

int main() {
  using namespace gogp;
  const int mt = 64, nt = 64;  // 4096 tiles = 8 rounds of 512
  const long M = (long)mt * 128, N = (long)nt * 128, KMAX = 2048;
  double *A, *B, *C, *C2;
  CK(hipMalloc(&A, M * KMAX * 8));
  CK(hipMalloc(&B, N * KMAX * 8));
  CK(hipMalloc(&C, M * N * 8));
  CK(hipMalloc(&C2, M * N * 8));
  std::vector<double> h((size_t)M * KMAX);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int K : {256, 512, 768, 1024, 2048}) {
    GemmArgs g{};
    g.A = A; g.B = B; g.lda = KMAX; g.ldb = KMAX; g.ldc = N; g.mt = mt; g.nt = nt; g.nkt = K / 16;
    g.alpha = -1.0; g.beta = 1.0; g.kend = K; g.Pr = g.Pc = 1;
    g.new_row0 = g.krag0 = 0x7fffffff; g.bstride = 0;
    const int ntiles = mt * nt, reps = 20;
    float ms_base = 0, ms_p[3] = {0, 0, 0};
    // correctness: one launch each from the same C
    CK(hipMemset(C, 0, M * N * 8));
    CK(hipMemset(C2, 0, M * N * 8));
    g.C = C;
    launch_dgemm_nt(s, GEMM_RECT, mt, nt, K, -1.0, A, KMAX, B, KMAX, 1.0, C, N, nullptr, nullptr);
    g.C = C2;
    hipLaunchKernelGGL((dgemm_persist_kernel<2>), dim3(512), dim3(512), 0, s, g, ntiles);
    CK(hipStreamSynchronize(s));
    std::vector<double> c1(4096), c2(4096);
    CK(hipMemcpy(c1.data(), C + 12345 * 3, 4096 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c2.data(), C2 + 12345 * 3, 4096 * 8, hipMemcpyDeviceToHost));
    bool same = true;
    for (int i = 0; i < 4096; ++i) same = same && c1[i] == c2[i];
    g.C = C;
    for (int w = 0; w < 3; ++w) launch_dgemm_nt(s, GEMM_RECT, mt, nt, K, -1.0, A, KMAX, B, KMAX, 1.0, C, N, nullptr, nullptr);
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) launch_dgemm_nt(s, GEMM_RECT, mt, nt, K, -1.0, A, KMAX, B, KMAX, 1.0, C, N, nullptr, nullptr);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_base, e0, e1));
    const int grids[3] = {512, 504, 1024};
    for (int v = 0; v < 3; ++v) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((dgemm_persist_kernel<2>), dim3(grids[v]), dim3(512), 0, s, g, ntiles);
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((dgemm_persist_kernel<2>), dim3(grids[v]), dim3(512), 0, s, g, ntiles);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_p[v], e0, e1));
    }
    const double fl = 2.0 * M * N * K;
    printf("K %5d  identical %d  product kernel %.4f ms %.2f TFLOP/s | persistent grid 512: %.4f ms %.2f | 504: %.4f ms %.2f | 1024: %.4f ms %.2f\n",
           K, (int)same, ms_base / reps, fl / (ms_base / reps * 1e-3) / 1e12, ms_p[0] / reps, fl / (ms_p[0] / reps * 1e-3) / 1e12,
           ms_p[1] / reps, fl / (ms_p[1] / reps * 1e-3) / 1e12, ms_p[2] / reps, fl / (ms_p[2] / reps * 1e-3) / 1e12);
  }
  return 0;
}
