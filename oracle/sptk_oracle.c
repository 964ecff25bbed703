/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see sptk_oracle_impl.h for the full header).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
 * library built from this file.  The product package (diffsptk_amd/) never does.
 */
#define _USE_MATH_DEFINES
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

#define REAL float
#define SFX(x) x##_f32
#include "sptk_oracle_impl.h"
#undef REAL
#undef SFX

#define REAL double
#define SFX(x) x##_f64
#include "sptk_oracle_impl.h"
#undef REAL
#undef SFX

/* modified Bessel function of the first kind, order 0 (power series) */
static double bessel_i0(double x)
{
    double s = 1, t = 1, q = x * x / 4;
    for (int k = 1; k < 200; ++k) {
        t *= q / ((double)k * (double)k);
        s += t;
        if (t < 1e-18 * s) break;
    }
    return s;
}

/*
 * Window._precompute, window.py:122-183: window table in double.  The reference calls
 * torch.{blackman,hamming,hann,bartlett,kaiser}_window and torch.signal.windows.cosine;
 * their published definitions are restated here.
 *   type: 0 blackman 1 hamming 2 hanning 3 bartlett 4 trapezoidal 5 rectangular 6 nuttall
 *         7 povey 8 sine 9 vorbis 10 kbd          norm: 0 none 1 power 2 magnitude
 * Returns 0, or -1 for an unsupported combination (kbd + periodic, window.py:164-165).
 */
EXPORT int oracle_window_table(int type, int L, int norm, int symmetric, double* w)
{
    int periodic = !symmetric;
    double D = periodic ? (double)L : (double)(L - 1); /* cosine-sum denominator */
    for (int n = 0; n < L; ++n) {
        double ph = (L == 1) ? 0.0 : 2.0 * M_PI * (double)n / D;
        double bart = (L == 1) ? 1.0 : 1.0 - fabs(2.0 * (double)n / D - 1.0);
        double hann = (L == 1) ? 1.0 : 0.5 - 0.5 * cos(ph);
        double cosw = sin(M_PI * ((double)n + 0.5) / (double)(symmetric ? L : L + 1));
        switch (type) {
        case 0: w[n] = (L == 1) ? 1.0 : 0.42 - 0.5 * cos(ph) + 0.08 * cos(2 * ph); break;
        case 1: w[n] = (L == 1) ? 1.0 : 0.54 - 0.46 * cos(ph); break;
        case 2: w[n] = hann; break;
        case 3: w[n] = bart; break;
        case 4: w[n] = fmin(2.0 * bart, 1.0); break;
        case 5: w[n] = 1.0; break;
        case 6: { /* nuttall, window.py:149-154 */
            double size = periodic ? (double)L : (double)(L - 1);
            double c1[4] = {0.355768, -0.487396, 0.144232, -0.012604};
            double s = 0;
            for (int k = 0; k < 4; ++k) s += c1[k] * cos((double)n * (2.0 * k) * (M_PI / size));
            w[n] = s;
            break;
        }
        case 7: w[n] = pow(hann, 0.85); break;
        case 8: w[n] = cosw; break;
        case 9: w[n] = sin(M_PI * 0.5 * cosw * cosw); break;
        case 10: break;
        default: return -1;
        }
    }
    if (type == 10) { /* kbd, window.py:163-169 */
        if (periodic) return -1;
        int Nk = L / 2 + 1;
        double* cs = (double*)malloc(sizeof(double) * Nk);
        double acc = 0;
        for (int n = 0; n < Nk; ++n) {
            double r = (Nk == 1) ? 0.0 : ((double)n - (Nk - 1) / 2.0) / ((Nk - 1) / 2.0);
            double k = (Nk == 1) ? 1.0 : bessel_i0(12.0 * sqrt(fmax(0.0, 1.0 - r * r))) / bessel_i0(12.0);
            acc += k;
            cs[n] = acc;
        }
        int half = Nk - 1;
        for (int n = 0; n < half; ++n) {
            double v = sqrt(cs[n] / cs[Nk - 1]);
            w[n] = v;
            w[2 * half - 1 - n] = v;
        }
        free(cs);
    }
    if (norm == 1) {
        double s = 0;
        for (int n = 0; n < L; ++n) s += w[n] * w[n];
        s = sqrt(s);
        for (int n = 0; n < L; ++n) w[n] /= s;
    } else if (norm == 2) {
        double s = 0;
        for (int n = 0; n < L; ++n) s += w[n];
        for (int n = 0; n < L; ++n) w[n] /= s;
    } else if (norm != 0) {
        return -1;
    }
    return 0;
}
