/* Threaded 3-D r2c / c2r transform for the CPU stand-in of the reference's loop (bench.py cpu_baseline) -- in place of the
 * threaded FFTW the reference links (fftw_plan_many_dft_r2c / _c2r with fftw_plan_with_nthreads, F:7130-7262) and of the
 * pocketfft workers round 2-4 used: plain C + OpenMP, so that OMP_PROC_BIND / OMP_PLACES pin EVERY thread of a pass and the
 * pages a thread transforms are the pages it first touched.
 *
 * TEST INFRASTRUCTURE ONLY: nothing under fibergen_amd/ links or calls this.
 *
 * Algorithm: the textbook row-column decomposition FFTW itself uses for a 3-D r2c -- 1-D real transforms along z (packed as
 * complex lines of nz/2 points + split), complex transforms along y, complex transforms along x -- each 1-D transform a
 * Stockham autosort radix-4 / radix-2 FFT on a batch of B lines held as structure-of-arrays [point][line] so that the
 * compiler vectorises across the lines.  Powers of two only (the BASELINE grids; ref_fft_supported says so, callers fall
 * back to pocketfft otherwise).  Unnormalised like FFTW (the 1/N sweep is the caller's, F:18501-18506).
 *
 * Layouts: real [ncomp][nx][ny][nz] (no padding), complex [ncomp][nx][ny][nz/2+1] interleaved (re, im).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define FB 16 /* lines per batch: two AVX-512 vectors of doubles */

static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

int ref_fft_supported(int nx, int ny, int nz) { return is_pow2(nx) && is_pow2(ny) && is_pow2(nz) && nz >= 4; }

/* w[k] = exp(-2 pi i k / n), k < n */
static void make_roots(int n, double* wr, double* wi) {
  for (int k = 0; k < n; k++) {
    const long double a = -6.283185307179586476925286766559005768L * (long double)k / (long double)n;
    wr[k] = (double)cosl(a);
    wi[k] = (double)sinl(a);
  }
}

/* Stockham autosort FFT of n points on FB lines, x = [n][FB] split re / im, y = scratch of the same size; the result ends in
 * x.  sign = -1 forward, +1 inverse (unnormalised).  wr / wi: roots of order n. */
static void fft_lines(int n, double* restrict xr, double* restrict xi, double* restrict yr, double* restrict yi,
                      const double* wr, const double* wi, int sign) {
  double *ar = xr, *ai = xi, *br = yr, *bi = yi;
  int ns = 1;
  /* radix-4 stages while four divides what is left */
  while (ns * 4 <= n && (n / ns) % 4 == 0) {
    const int q = n / 4;
    for (int j = 0; j < q; j++) {
      const int k = j % ns;
      const int t = k * (n / (4 * ns));
      const double w1r = wr[t], w1i = sign < 0 ? wi[t] : -wi[t];
      const double w2r = wr[2 * t], w2i = sign < 0 ? wi[2 * t] : -wi[2 * t];
      const double w3r = wr[3 * t], w3i = sign < 0 ? wi[3 * t] : -wi[3 * t];
      const double* p0r = ar + (size_t)j * FB;
      const double* p0i = ai + (size_t)j * FB;
      const double* p1r = p0r + (size_t)q * FB;
      const double* p1i = p0i + (size_t)q * FB;
      const double* p2r = p1r + (size_t)q * FB;
      const double* p2i = p1i + (size_t)q * FB;
      const double* p3r = p2r + (size_t)q * FB;
      const double* p3i = p2i + (size_t)q * FB;
      const size_t o = ((size_t)(j / ns) * 4 * ns + k) * FB;
      double* o0r = br + o;
      double* o0i = bi + o;
      double* o1r = o0r + (size_t)ns * FB;
      double* o1i = o0i + (size_t)ns * FB;
      double* o2r = o1r + (size_t)ns * FB;
      double* o2i = o1i + (size_t)ns * FB;
      double* o3r = o2r + (size_t)ns * FB;
      double* o3i = o2i + (size_t)ns * FB;
#pragma omp simd
      for (int b = 0; b < FB; b++) {
        const double a0r = p0r[b], a0i = p0i[b];
        const double a1r = p1r[b] * w1r - p1i[b] * w1i, a1i = p1r[b] * w1i + p1i[b] * w1r;
        const double a2r = p2r[b] * w2r - p2i[b] * w2i, a2i = p2r[b] * w2i + p2i[b] * w2r;
        const double a3r = p3r[b] * w3r - p3i[b] * w3i, a3i = p3r[b] * w3i + p3i[b] * w3r;
        const double s02r = a0r + a2r, s02i = a0i + a2i, d02r = a0r - a2r, d02i = a0i - a2i;
        const double s13r = a1r + a3r, s13i = a1i + a3i, d13r = a1r - a3r, d13i = a1i - a3i;
        /* forward: multiply the odd difference by -i, inverse: by +i */
        const double jr = sign < 0 ? d13i : -d13i, ji = sign < 0 ? -d13r : d13r;
        o0r[b] = s02r + s13r; o0i[b] = s02i + s13i;
        o1r[b] = d02r + jr;   o1i[b] = d02i + ji;
        o2r[b] = s02r - s13r; o2i[b] = s02i - s13i;
        o3r[b] = d02r - jr;   o3i[b] = d02i - ji;
      }
    }
    double* t;
    t = ar; ar = br; br = t;
    t = ai; ai = bi; bi = t;
    ns *= 4;
  }
  while (ns < n) { /* one radix-2 stage when log2 n is odd */
    const int h = n / 2;
    for (int j = 0; j < h; j++) {
      const int k = j % ns;
      const int t = k * (n / (2 * ns));
      const double w1r = wr[t], w1i = sign < 0 ? wi[t] : -wi[t];
      const double* p0r = ar + (size_t)j * FB;
      const double* p0i = ai + (size_t)j * FB;
      const double* p1r = p0r + (size_t)h * FB;
      const double* p1i = p0i + (size_t)h * FB;
      const size_t o = ((size_t)(j / ns) * 2 * ns + k) * FB;
      double* o0r = br + o;
      double* o0i = bi + o;
      double* o1r = o0r + (size_t)ns * FB;
      double* o1i = o0i + (size_t)ns * FB;
#pragma omp simd
      for (int b = 0; b < FB; b++) {
        const double a0r = p0r[b], a0i = p0i[b];
        const double a1r = p1r[b] * w1r - p1i[b] * w1i, a1i = p1r[b] * w1i + p1i[b] * w1r;
        o0r[b] = a0r + a1r; o0i[b] = a0i + a1i;
        o1r[b] = a0r - a1r; o1i[b] = a0i - a1i;
      }
    }
    double* t;
    t = ar; ar = br; br = t;
    t = ai; ai = bi; bi = t;
    ns *= 2;
  }
  if (ar != xr) {
    memcpy(xr, ar, (size_t)n * FB * sizeof(double));
    memcpy(xi, ai, (size_t)n * FB * sizeof(double));
  }
}

typedef struct {
  int nx, ny, nz, nzc;
  double *wxr, *wxi, *wyr, *wyi, *wmr, *wmi, *wzr, *wzi; /* roots of order nx, ny, nz/2, nz */
} fft_plan;

void* ref_fft_plan(int nx, int ny, int nz) {
  if (!ref_fft_supported(nx, ny, nz)) return NULL;
  fft_plan* p = (fft_plan*)calloc(1, sizeof(fft_plan));
  p->nx = nx; p->ny = ny; p->nz = nz; p->nzc = nz / 2 + 1;
  p->wxr = (double*)malloc(sizeof(double) * nx); p->wxi = (double*)malloc(sizeof(double) * nx);
  p->wyr = (double*)malloc(sizeof(double) * ny); p->wyi = (double*)malloc(sizeof(double) * ny);
  p->wmr = (double*)malloc(sizeof(double) * (nz / 2)); p->wmi = (double*)malloc(sizeof(double) * (nz / 2));
  p->wzr = (double*)malloc(sizeof(double) * nz); p->wzi = (double*)malloc(sizeof(double) * nz);
  make_roots(nx, p->wxr, p->wxi);
  make_roots(ny, p->wyr, p->wyi);
  make_roots(nz / 2, p->wmr, p->wmi);
  make_roots(nz, p->wzr, p->wzi);
  return p;
}

void ref_fft_plan_free(void* plan) {
  fft_plan* p = (fft_plan*)plan;
  if (!p) return;
  free(p->wxr); free(p->wxi); free(p->wyr); free(p->wyi); free(p->wmr); free(p->wmi); free(p->wzr); free(p->wzi);
  free(p);
}

/* complex transform along a strided axis of one x-plane / y-row set: lines of n points `ls` complex apart, FB adjacent
 * columns at a time starting at complex offset base + c0 */
static void strided_lines(int n, double* data, size_t ls, int ncols_here, double* buf, const double* wr, const double* wi,
                          int sign) {
  double *xr = buf, *xi = buf + (size_t)n * FB, *yr = xi + (size_t)n * FB, *yi = yr + (size_t)n * FB;
  for (int j = 0; j < n; j++) {
    const double* src = data + 2 * (size_t)j * ls;
    for (int b = 0; b < FB; b++) {
      const int bb = b < ncols_here ? b : ncols_here - 1; /* ragged last batch: duplicate the last column */
      xr[(size_t)j * FB + b] = src[2 * bb];
      xi[(size_t)j * FB + b] = src[2 * bb + 1];
    }
  }
  fft_lines(n, xr, xi, yr, yi, wr, wi, sign);
  for (int j = 0; j < n; j++) {
    double* dst = data + 2 * (size_t)j * ls;
    for (int b = 0; b < ncols_here; b++) {
      dst[2 * b] = xr[(size_t)j * FB + b];
      dst[2 * b + 1] = xi[(size_t)j * FB + b];
    }
  }
}

/* f [ncomp][nx][ny][nz] real -> fh [ncomp][nx][ny][nzc] complex, unnormalised (FFTW r2c semantics, F:18496-18499) */
void ref_fft_r2c(void* plan, int ncomp, const double* f, double* fh) {
  const fft_plan* p = (const fft_plan*)plan;
  const int nx = p->nx, ny = p->ny, nz = p->nz, nzc = p->nzc, M = nz / 2;
  const size_t nmax = (size_t)(nx > ny ? nx : ny) > (size_t)M ? (size_t)(nx > ny ? nx : ny) : (size_t)M;
#pragma omp parallel
  {
    double* buf = (double*)malloc(sizeof(double) * 4 * nmax * FB);
    double *xr = buf, *xi = buf + (size_t)M * FB, *yr = xi + (size_t)M * FB, *yi = yr + (size_t)M * FB;
    /* z: rows of nz reals packed as M complex points, FB rows at a time, then the real split; y right after on the
     * same x-plane (it is still in this thread's cache hierarchy) */
    /* (per component a static loop over the x-planes: thread t owns the same planes of every component and of every
     * field the loop nests of fg_ref.c sweep with their static (x, y) schedules -- the pages ref_first_touch placed) */
    for (int c = 0; c < ncomp; c++)
#pragma omp for schedule(static) nowait
      for (int i = 0; i < nx; i++) {
        const double* src = f + ((size_t)c * nx + i) * ny * nz;
        double* dst = fh + 2 * ((size_t)c * nx + i) * ny * nzc;
        for (int j0 = 0; j0 < ny; j0 += FB) {
          const int nb = ny - j0 < FB ? ny - j0 : FB;
          for (int b = 0; b < FB; b++) {
            const double* row = src + (size_t)(j0 + (b < nb ? b : nb - 1)) * nz;
            for (int m = 0; m < M; m++) {
              xr[(size_t)m * FB + b] = row[2 * m];
              xi[(size_t)m * FB + b] = row[2 * m + 1];
            }
          }
          fft_lines(M, xr, xi, yr, yi, p->wmr, p->wmi, -1);
          for (int b = 0; b < nb; b++) {
            double* out = dst + 2 * (size_t)(j0 + b) * nzc;
            for (int k = 0; k <= M; k++) {
              const int k1 = k == M ? 0 : k, k2 = k == 0 ? 0 : M - k;
              const double zr = xr[(size_t)k1 * FB + b], zi = xi[(size_t)k1 * FB + b];
              const double mr = xr[(size_t)k2 * FB + b], mi = -xi[(size_t)k2 * FB + b]; /* conj Z[M-k] */
              const double er = 0.5 * (zr + mr), ei = 0.5 * (zi + mi);
              const double dr = 0.5 * (zr - mr), di = 0.5 * (zi - mi);
              const double or_ = p->wzr[k] * dr - p->wzi[k] * di, oi = p->wzr[k] * di + p->wzi[k] * dr;
              out[2 * k] = er + oi;     /* e - i o */
              out[2 * k + 1] = ei - or_;
            }
          }
        }
        for (int k0 = 0; k0 < nzc; k0 += FB)
          strided_lines(ny, dst + 2 * (size_t)k0, (size_t)nzc, nzc - k0 < FB ? nzc - k0 : FB, buf, p->wyr, p->wyi, -1);
      }
#pragma omp barrier
    for (int c = 0; c < ncomp; c++)
#pragma omp for schedule(static) nowait
      for (int j = 0; j < ny; j++) {
        double* base = fh + 2 * (((size_t)c * nx) * ny + j) * nzc;
        for (int k0 = 0; k0 < nzc; k0 += FB)
          strided_lines(nx, base + 2 * (size_t)k0, (size_t)ny * nzc, nzc - k0 < FB ? nzc - k0 : FB, buf, p->wxr, p->wxi, -1);
      }
    free(buf);
  }
}

/* fh [ncomp][nx][ny][nzc] complex -> u [ncomp][nx][ny][nz] real, unnormalised (FFTW c2r semantics, F:18513-18530); fh is
 * overwritten like FFTW's c2r input */
void ref_fft_c2r(void* plan, int ncomp, double* fh, double* u) {
  const fft_plan* p = (const fft_plan*)plan;
  const int nx = p->nx, ny = p->ny, nz = p->nz, nzc = p->nzc, M = nz / 2;
  const size_t nmax = (size_t)(nx > ny ? nx : ny) > (size_t)M ? (size_t)(nx > ny ? nx : ny) : (size_t)M;
#pragma omp parallel
  {
    double* buf = (double*)malloc(sizeof(double) * 4 * nmax * FB);
    double *xr = buf, *xi = buf + (size_t)M * FB, *yr = xi + (size_t)M * FB, *yi = yr + (size_t)M * FB;
    for (int c = 0; c < ncomp; c++)
#pragma omp for schedule(static) nowait
      for (int j = 0; j < ny; j++) {
        double* base = fh + 2 * (((size_t)c * nx) * ny + j) * nzc;
        for (int k0 = 0; k0 < nzc; k0 += FB)
          strided_lines(nx, base + 2 * (size_t)k0, (size_t)ny * nzc, nzc - k0 < FB ? nzc - k0 : FB, buf, p->wxr, p->wxi, +1);
      }
#pragma omp barrier
    for (int c = 0; c < ncomp; c++)
#pragma omp for schedule(static) nowait
      for (int i = 0; i < nx; i++) {
        double* src = fh + 2 * ((size_t)c * nx + i) * ny * nzc;
        double* dst = u + ((size_t)c * nx + i) * ny * nz;
        for (int k0 = 0; k0 < nzc; k0 += FB)
          strided_lines(ny, src + 2 * (size_t)k0, (size_t)nzc, nzc - k0 < FB ? nzc - k0 : FB, buf, p->wyr, p->wyi, +1);
        for (int j0 = 0; j0 < ny; j0 += FB) {
          const int nb = ny - j0 < FB ? ny - j0 : FB;
          for (int b = 0; b < FB; b++) {
            const double* in = src + 2 * (size_t)(j0 + (b < nb ? b : nb - 1)) * nzc;
            for (int k = 0; k < M; k++) {
              /* Z'[k] = (X[k] + conj X[M-k]) + i conj(w^k) (X[k] - conj X[M-k]) */
              /* (k = 0 pairs the DC and the Nyquist bin, whose imaginary parts a c2r transform ignores) */
              const double ar = in[2 * k], ai = k ? in[2 * k + 1] : 0.0;
              const double mr = in[2 * (M - k)], mi = k ? -in[2 * (M - k) + 1] : 0.0;
              const double er = ar + mr, ei = ai + mi, dr = ar - mr, di = ai - mi;
              const double or_ = p->wzr[k] * dr + p->wzi[k] * di, oi = p->wzr[k] * di - p->wzi[k] * dr; /* conj(w) d */
              xr[(size_t)k * FB + b] = er - oi; /* e + i o */
              xi[(size_t)k * FB + b] = ei + or_;
            }
          }
          fft_lines(M, xr, xi, yr, yi, p->wmr, p->wmi, +1);
          for (int b = 0; b < nb; b++) {
            double* row = dst + (size_t)(j0 + b) * nz;
            for (int m = 0; m < M; m++) {
              row[2 * m] = xr[(size_t)m * FB + b];
              row[2 * m + 1] = xi[(size_t)m * FB + b];
            }
          }
        }
      }
    free(buf);
  }
}

/* Zero-fill of a field [ncomp][nx][plane] by the threads that will work on it: per component a static loop over the x-planes
 * (the schedule of the transforms above and of the static (x, y) loop nests), so that first touch places every page on the
 * NUMA node of its thread. */
void ref_first_touch(double* a, int ncomp, int nx, size_t plane) {
#pragma omp parallel
  for (int c = 0; c < ncomp; c++)
#pragma omp for schedule(static) nowait
    for (int i = 0; i < nx; i++) memset(a + ((size_t)c * nx + i) * plane, 0, plane * sizeof(double));
}
