/* The drop-in boundary from plain C: one load case of a two-phase cell through include/fibergen_amd.h -- what a binding inside
 * fibergen (C++) or any other host language does.
 *
 *   gcc -std=c99 -Iinclude examples/c_abi_example.c -Lfibergen_amd -lfibergen_amd -Wl,-rpath,$PWD/fibergen_amd -lm -o /tmp/c_abi_example
 *   /tmp/c_abi_example 32
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "fibergen_amd.h"

#define CHECK(call)                                                       \
  do {                                                                    \
    if ((call) != 0) {                                                    \
      fprintf(stderr, "%s failed: %s\n", #call, fg_last_error(s));       \
      return 1;                                                           \
    }                                                                     \
  } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 32;
  const size_t N = (size_t)n * n * n;
  double* phi0 = (double*)malloc(N * sizeof(double));
  double* phi1 = (double*)malloc(N * sizeof(double));
  if (!phi0 || !phi1) return 2;
  /* a centred sphere of radius 0.3 (binary fractions, voxel centres) */
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      for (int k = 0; k < n; ++k) {
        const double x = (i + 0.5) / n - 0.5, y = (j + 0.5) / n - 0.5, z = (k + 0.5) / n - 0.5;
        const double in = sqrt(x * x + y * y + z * z) < 0.3 ? 1.0 : 0.0;
        phi1[((size_t)i * n + j) * n + k] = in;
        phi0[((size_t)i * n + j) * n + k] = 1.0 - in;
      }
  if (fg_abi_version() != 1) return 3;
  fg_solver* s = fg_create(n, n, n, 1.0, 1.0, 1.0, 0);
  if (!s) {
    fprintf(stderr, "fg_create: %s\n", fg_last_error(NULL));
    return 4;
  }
  /* Lame constants of E = 1, nu = 0.3 (matrix) and E = 10, nu = 0.2 (inclusion) */
  const double mu0 = 1.0 / (2 * 1.3), la0 = 0.3 / (1.3 * 0.4), mu1 = 10.0 / (2 * 1.2), la1 = 10.0 * 0.2 / (1.2 * 0.6);
  CHECK(fg_set_num_phases(s, 2));
  CHECK(fg_set_phase(s, 0, mu0, la0, phi0));
  CHECK(fg_set_phase(s, 1, mu1, la1, phi1));
  CHECK(fg_set_option_d(s, "tol", 1e-8));
  CHECK(fg_set_option_i(s, "method", 1)); /* conjugate gradients, the reference's default */
  const double E[6] = {0.01, 0, 0, 0, 0, 0}, S[6] = {0, 0, 0, 0, 0, 0};
  int failed = 0;
  CHECK(fg_run_load_case(s, E, S, &failed));
  double sig[6], vf = 0.0;
  CHECK(fg_mean_stress(s, sig));
  CHECK(fg_volume_fraction(s, 1, &vf));
  printf("n = %d, inclusion fraction %.6f, %ld iterations%s, solve time %.3f ms\n", n, vf, fg_get_iterations(s),
         failed ? " (FAILED)" : "", 1e3 * fg_get_solve_time(s));
  printf("mean stress: %.9e %.9e %.9e %.9e %.9e %.9e\n", sig[0], sig[1], sig[2], sig[3], sig[4], sig[5]);
  fg_destroy(s);
  free(phi0);
  free(phi1);
  return failed ? 5 : 0;
}
