// Host-side twiddle tables (computed in long double, rounded once to double).
#pragma once

#include <cmath>
#include <vector>

#include "fg_fft_core.h"

namespace fg {

// e^{-2 pi i k/n}, k = 0..count-1
inline std::vector<cplx> make_unit_roots(int n, int count) {
  std::vector<cplx> w(count);
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int k = 0; k < count; ++k) {
    // reduce the angle to the first octant for full accuracy
    long double a = two_pi * (long double)(k % n) / (long double)n;
    w[k] = cmake((double)cosl(a), (double)-sinl(a));
  }
  return w;
}

// Pass twiddles of a power-of-two N: [pass>=1][(r-1)*N/R + j] = e^{-2 pi i r (j % Ns)/(Ns R)}
inline std::vector<cplx> make_pass_twiddles(int N) {
  using namespace fft;
  std::vector<cplx> tw;
  if (!is_pow2(N) || N < 8) return tw;
  const long double two_pi = 6.283185307179586476925286766559005768L;
  int np = num_passes(N);
  int ns = 1;
  for (int p = 0; p < np; ++p) {
    int R = pass_radix(N, p);
    if (p >= 1) {
      int nbf = N / R;
      for (int r = 1; r < R; ++r)
        for (int j = 0; j < nbf; ++j) {
          long double a = two_pi * (long double)(r * (j % ns)) / (long double)(ns * R);
          tw.push_back(cmake((double)cosl(a), (double)-sinl(a)));
        }
    }
    ns *= R;
  }
  if (tw.empty()) tw.push_back(cmake(1.0, 0.0));
  return tw;
}

}  // namespace fg
