// Census of where workgroups land: HW_REG_HW_ID (id 4) and HW_REG_XCC_ID (id 20).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void census(unsigned *out, int spin) {
  unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // size-1=31, offset 0, id 4
  unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
  // keep the CU busy for a while so that the grid spreads over all CUs
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
  const int nb = 4096;
  unsigned *d; hipMalloc(&d, nb * 8);
  hipLaunchKernelGGL(census, dim3(nb), dim3(256), 65536, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * nb);
  hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
  std::set<unsigned long long> full;
  std::map<unsigned, int> f_cu, f_sh, f_se, f_xcc;
  for (int i = 0; i < nb; ++i) {
    unsigned hw = h[2 * i], xcc = h[2 * i + 1];
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, x = xcc & 0xf;
    f_cu[cu]++; f_sh[sh]++; f_se[se]++; f_xcc[x]++;
    full.insert(((unsigned long long)x << 32) | (se << 8) | (sh << 4) | cu);
  }
  printf("distinct (xcc,se,sh,cu) tuples: %zu\n", full.size());
  printf("cu ids:"); for (auto &p : f_cu) printf(" %u:%d", p.first, p.second); printf("\n");
  printf("sh ids:"); for (auto &p : f_sh) printf(" %u:%d", p.first, p.second); printf("\n");
  printf("se ids:"); for (auto &p : f_se) printf(" %u:%d", p.first, p.second); printf("\n");
  printf("xcc ids:"); for (auto &p : f_xcc) printf(" %u:%d", p.first, p.second); printf("\n");
  printf("raw samples: %08x/%x %08x/%x %08x/%x\n", h[0], h[1], h[2], h[3], h[200], h[201]);
  int n0 = 0; for (int i = 0; i < nb; ++i) { unsigned hw = h[2*i]; if (((hw >> 8) & 0xf) == 0 && ((hw >> 12) & 1) == 0 && ((hw >> 13) & 7) == 0 && (h[2*i+1] & 0xf) == 0) n0++; }
  printf("blocks on (xcc0,se0,sh0,cu0): %d\n", n0);
  return 0;
}
