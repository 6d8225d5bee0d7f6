/* ORACLE (test infrastructure, not product code).
 *
 * Sequential selective scan, fp32, one recurrence per (batch, channel, state).
 * Restates the semantics of the third-party op the reference calls at
 * /root/reference/src/emamba2.py:154  selective_scan_cuda_core.fwd(u, delta, A, B, C, D,
 * delta_bias, delta_softplus, nrows)  (VMamba kernels/selective_scan; NOT vendored and NOT
 * version-pinned by the reference: install.yaml has no entry) following the published
 * `selective_scan_ref`:
 *     dt   = softplus(delta + delta_bias[d])              (torch softplus, threshold 20)
 *     h_t  = exp(dt * A[d,n]) * h_{t-1} + dt * B[b,g,n,t] * u[b,d,t]
 *     y_t  = sum_n h_t * C[b,g,n,t] + D[d] * u[b,d,t]      g = d / (KD / K)
 * PARITY UNPINNED for this op: the reference holds no test or golden vector for it; the
 * only anchor is the call site above and the shapes documented at emamba2.py:38-51.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC (see oracle/build.py).
 */
#include <math.h>
#include <stdlib.h>

static inline float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

void fd_oracle_selective_scan(const float *u, const float *delta, const float *A, const float *Bm,
                              const float *Cm, const float *D, const float *dbias, int softplus,
                              float *y, int b, int KD, int K, int N, long L)
{
    const int Dg = KD / K;
#pragma omp parallel for collapse(2) schedule(static)
    for (int ib = 0; ib < b; ++ib) {
        for (int d = 0; d < KD; ++d) {
            const int g = d / Dg;
            const float *ur = u + ((long)ib * KD + d) * L;
            const float *dr = delta + ((long)ib * KD + d) * L;
            float *yr = y + ((long)ib * KD + d) * L;
            const float *Br = Bm + ((long)ib * K + g) * N * L;
            const float *Cr = Cm + ((long)ib * K + g) * N * L;
            const float *Ar = A + (long)d * N;
            float h[256];
            for (int n = 0; n < N; ++n) h[n] = 0.0f;
            const float bias = dbias ? dbias[d] : 0.0f;
            const float Dd = D ? D[d] : 0.0f;
            for (long t = 0; t < L; ++t) {
                float dt = dr[t] + bias;
                if (softplus) dt = softplus_f(dt);
                const float ut = ur[t];
                float acc = 0.0f;
                for (int n = 0; n < N; ++n) {
                    const float dA = expf(dt * Ar[n]);
                    h[n] = dA * h[n] + dt * Br[(long)n * L + t] * ut;
                    acc += h[n] * Cr[(long)n * L + t];
                }
                yr[t] = acc + Dd * ut;
            }
        }
    }
}

/* The same recurrence in double precision (inputs fp32, state and arithmetic fp64): the yardstick for the
 * long-sequence case (one L = 65 536 case against fp64). */
void fd_oracle_selective_scan_f64(const float *u, const float *delta, const float *A, const float *Bm,
                                  const float *Cm, const float *D, const float *dbias, int softplus,
                                  double *y, int b, int KD, int K, int N, long L)
{
    const int Dg = KD / K;
#pragma omp parallel for collapse(2) schedule(static)
    for (int ib = 0; ib < b; ++ib) {
        for (int d = 0; d < KD; ++d) {
            const int g = d / Dg;
            const float *ur = u + ((long)ib * KD + d) * L;
            const float *dr = delta + ((long)ib * KD + d) * L;
            double *yr = y + ((long)ib * KD + d) * L;
            const float *Br = Bm + ((long)ib * K + g) * N * L;
            const float *Cr = Cm + ((long)ib * K + g) * N * L;
            const float *Ar = A + (long)d * N;
            double h[256];
            for (int n = 0; n < N; ++n) h[n] = 0.0;
            const double bias = dbias ? dbias[d] : 0.0;
            const double Dd = D ? D[d] : 0.0;
            for (long t = 0; t < L; ++t) {
                double dt = (double)dr[t] + bias;
                if (softplus) dt = dt > 20.0 ? dt : log1p(exp(dt));
                const double ut = ur[t];
                double acc = 0.0;
                for (int n = 0; n < N; ++n) {
                    h[n] = exp(dt * (double)Ar[n]) * h[n] + dt * (double)Br[(long)n * L + t] * ut;
                    acc += h[n] * (double)Cr[(long)n * L + t];
                }
                yr[t] = acc + Dd * ut;
            }
        }
    }
}
