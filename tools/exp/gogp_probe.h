/* gogp_probe.h -- forensic probe of the gradient-reduction instances removed in round 2.
 * NOT part of the product or of the default hook library: `make -C gogp_amd/csrc probe` builds
 * gogp_amd/libgogp_probe.so from tools/exp/grad64_probe.hip + gradold_probe.hip (the pre-round-2 source kept verbatim
 * under tools/exp/gradold/); tools/agpr_probe.py drives it. */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic for the gradient-reduction instances removed in round 2: rebuilds the
 * 64-accumulator LOCAL instance of the per-pair kernel in the hook library and runs it on random data
 * (n rows, D in 33..64 ARD dimensions, 1 x 1 grid): out receives NACC slot sums per run -- run 0 the kept
 * instances of the product library (reference), runs 1..reps the 64-accumulator instance as it is, then two
 * runs per scrub value, each right behind a kernel that has filled every VGPR and AGPR of every SIMD with that
 * 32-bit pattern.  out: (1 + reps + 2 * nscrub) x 80 doubles. */
int gogp_test_grad64(int device, int64_t n, int D, int reps, const unsigned *scrub_values, int nscrub,
                     int variant /* 0: rebuilt from today's template; 1: the pre-round-2 source, sharded form (the one
                                    that failed); 2: its unsharded form; >= 16: the failed instance on that many
                                    workgroups (<= 256: each one is the first on its SIMDs after a scrub) */,
                     double *out);

#ifdef __cplusplus
}
#endif
