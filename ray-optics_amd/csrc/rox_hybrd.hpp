// rox_hybrd.hpp -- MINPACK `hybrd` (Powell's hybrid method) for a small compile-time n on the
// device: what scipy.optimize.fsolve runs for the 2-D branch of the reference's chief-ray
// aiming (rayoptics/raytr/trace.py:404-410 -> SciPy's _minpack: MINPACK-1 hybrd / fdjac1 /
// qrfac / qform / dogleg / r1updt / r1mpyq / enorm with mode = 1, a dense forward-difference
// Jacobian, nprint = 0).  One lane runs one problem; every array is a handful of registers.
// The operations and their order are the Fortran's (the CPU restatement the tests check this
// against is itself pinned bit for bit against the installed SciPy).  Every loop over the
// compile-time n is unrolled so that the arrays are registers, not scratch memory.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace rox {
namespace hybrd {

constexpr double EPSMCH = 2.220446049250313e-16;    // dpmpar(1)
constexpr double GIANT = 1.7976931348623157e308;    // dpmpar(3)

// enorm.f: Euclidean norm with separate sums for small / intermediate / large components
__device__ __forceinline__ double enorm(int n, const double *x)
{
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    const double agiant = rgiant / (double)n;
    #pragma unroll
    for (int i = 0; i < n; i++) {
        const double xabs = fabs(x[i]);
        if (xabs > rdwarf && xabs < agiant) {
            s2 += xabs * xabs;
        } else if (xabs <= rdwarf) {
            if (xabs > x3max) {
                const double q = x3max / xabs;
                s3 = 1.0 + s3 * (q * q);
                x3max = xabs;
            } else if (xabs != 0.0) {
                const double q = xabs / x3max;
                s3 += q * q;
            }
        } else {
            if (xabs > x1max) {
                const double q = x1max / xabs;
                s1 = 1.0 + s1 * (q * q);
                x1max = xabs;
            } else {
                const double q = xabs / x1max;
                s1 += q * q;
            }
        }
    }
    if (s1 != 0.0)
        return x1max * sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0.0) {
        if (s2 >= x3max)
            return sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
        return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrt(s3);
}

// Givens rotation of r1updt.f: eliminates `b` against `a`; tau is what r1mpyq recovers it from
__device__ __forceinline__ void givens(double a, double b, double &c, double &s, double &tau)
{
    if (fabs(a) < fabs(b)) {
        const double cotan = a / b;
        s = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
        c = s * cotan;
        tau = 1.0;
        if (fabs(c) * GIANT > 1.0)
            tau = 1.0 / c;
    } else {
        const double tn = b / a;
        c = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
        s = c * tn;
        tau = s;
    }
}

__device__ __forceinline__ void recover(double t, double &c, double &s)
{
    if (fabs(t) > 1.0) {
        c = 1.0 / t;
        s = sqrt(1.0 - c * c);
    } else {
        s = t;
        c = sqrt(1.0 - s * s);
    }
}

// hybrd.f.  `fcn(x, fvec)` returns false to stop the iteration (MINPACK's iflag < 0): solve then
// returns a negative info.  x holds the starting point and receives the last accepted iterate.
// `pts(x, h, need_f, fvec, cols)` evaluates the points of one forward-difference Jacobian --
// f(x) into fvec when need_f (hybrd's evaluation at the starting point), f(x + h[j] e_j) into
// cols + j * N -- in that order, and returns false as soon as one of them stops the iteration.
// fdjac1 evaluates them one after the other, but none depends on another's value: a caller with
// idle lanes traces them in one pass (rox_search.hpp), a caller without calls fcn in a loop.
template <int N, class F, class P>
__device__ __forceinline__ int solve(F &fcn, P &pts, double *x, double xtol, int maxfev,
                                     double epsfcn, double factor, int &nfev)
{
    constexpr int n = N;
#define FJ(i, j) fjac[(i) + (j) * n]
    double fvec[N], fjac[N * N], r[N * (N + 1) / 2], qtf[N];
    double diag[N], wa1[N], wa2[N], wa3[N], wa4[N];
    int info = 0;
    double xnorm = 0.0, delta = 0.0;
    bool ok;
    bool at_start = true;
    double fnorm = 0.0;
    nfev = 0;
    int iter = 1, ncsuc = 0, ncfail = 0, nslow1 = 0, nslow2 = 0;
    for (;;) {                                          // outer loop
        bool jeval = true;
        // fdjac1.f, dense: forward differences
        {
            const double eps = sqrt(epsfcn > EPSMCH ? epsfcn : EPSMCH);
            double h[N], cols[N * N];
            #pragma unroll
            for (int j = 0; j < n; j++) {
                h[j] = eps * fabs(x[j]);
                if (h[j] == 0.0)
                    h[j] = eps;
            }
            ok = pts(x, h, at_start, fvec, cols);
            if (at_start)
                nfev = 1;
            nfev += n;
            if (!ok)
                return -1;
            if (at_start) {
                fnorm = enorm(n, fvec);
                at_start = false;
            }
            #pragma unroll
            for (int j = 0; j < n; j++)
                #pragma unroll
                for (int i = 0; i < n; i++)
                    FJ(i, j) = (cols[j * n + i] - fvec[i]) / h[j];
        }
        // qrfac.f (no pivoting): rdiag = wa1, acnorm = wa2
        #pragma unroll
        for (int j = 0; j < n; j++) {
            wa2[j] = enorm(n, &FJ(0, j));
            wa1[j] = wa2[j];
        }
        #pragma unroll
        for (int j = 0; j < n; j++) {
            double ajnorm = enorm(n - j, &FJ(j, j));
            if (ajnorm != 0.0) {
                if (FJ(j, j) < 0.0)
                    ajnorm = -ajnorm;
                #pragma unroll
                for (int i = j; i < n; i++)
                    FJ(i, j) /= ajnorm;
                FJ(j, j) += 1.0;
                #pragma unroll
                for (int k = j + 1; k < n; k++) {
                    double sum = 0.0;
                    #pragma unroll
                    for (int i = j; i < n; i++)
                        sum += FJ(i, j) * FJ(i, k);
                    const double temp = sum / FJ(j, j);
                    #pragma unroll
                    for (int i = j; i < n; i++)
                        FJ(i, k) -= temp * FJ(i, j);
                }
            }
            wa1[j] = -ajnorm;
        }
        if (iter == 1) {
            #pragma unroll
            for (int j = 0; j < n; j++) {
                diag[j] = wa2[j];
                if (wa2[j] == 0.0)
                    diag[j] = 1.0;
            }
            #pragma unroll
            for (int j = 0; j < n; j++)
                wa3[j] = diag[j] * x[j];
            xnorm = enorm(n, wa3);
            delta = factor * xnorm;
            if (delta == 0.0)
                delta = factor;
        }
        // (q transpose) * fvec
        #pragma unroll
        for (int i = 0; i < n; i++)
            qtf[i] = fvec[i];
        #pragma unroll
        for (int j = 0; j < n; j++) {
            if (FJ(j, j) != 0.0) {
                double sum = 0.0;
                #pragma unroll
                for (int i = j; i < n; i++)
                    sum += FJ(i, j) * qtf[i];
                const double temp = -sum / FJ(j, j);
                #pragma unroll
                for (int i = j; i < n; i++)
                    qtf[i] += FJ(i, j) * temp;
            }
        }
        // the triangular factor, by rows
        #pragma unroll
        for (int j = 0; j < n; j++) {
            int l = j;
            #pragma unroll
            for (int i = 0; i < j; i++) {
                r[l] = FJ(i, j);
                l += n - 1 - i;
            }
            r[l] = wa1[j];
        }
        // qform.f
        #pragma unroll
        for (int j = 1; j < n; j++)
            #pragma unroll
            for (int i = 0; i < j; i++)
                FJ(i, j) = 0.0;
        #pragma unroll
        for (int l = 0; l < n; l++) {
            const int k = n - 1 - l;
            #pragma unroll
            for (int i = k; i < n; i++) {
                wa1[i] = FJ(i, k);
                FJ(i, k) = 0.0;
            }
            FJ(k, k) = 1.0;
            if (wa1[k] == 0.0)
                continue;
            #pragma unroll
            for (int j = k; j < n; j++) {
                double sum = 0.0;
                #pragma unroll
                for (int i = k; i < n; i++)
                    sum += FJ(i, j) * wa1[i];
                const double temp = sum / wa1[k];
                #pragma unroll
                for (int i = k; i < n; i++)
                    FJ(i, j) -= temp * wa1[i];
            }
        }
        #pragma unroll
        for (int j = 0; j < n; j++)
            diag[j] = diag[j] > wa2[j] ? diag[j] : wa2[j];
        for (;;) {                                      // inner loop
            // dogleg.f: direction into wa1 (x of dogleg), scratch wa2, wa3
            {
                int jj = (n * (n + 1)) / 2;
                #pragma unroll
                for (int k = 1; k <= n; k++) {
                    const int j = n - k;
                    jj -= k;
                    int l = jj + 1;
                    double sum = 0.0;
                    #pragma unroll
                    for (int i = j + 1; i < n; i++) {
                        sum += r[l] * wa1[i];
                        l++;
                    }
                    double temp = r[jj];
                    if (temp == 0.0) {
                        l = j;
                        #pragma unroll
                        for (int i = 0; i <= j; i++) {
                            const double t = fabs(r[l]);
                            if (t > temp)
                                temp = t;
                            l += n - 1 - i;
                        }
                        temp = EPSMCH * temp;
                        if (temp == 0.0)
                            temp = EPSMCH;
                    }
                    wa1[j] = (qtf[j] - sum) / temp;
                }
                #pragma unroll
                for (int j = 0; j < n; j++) {
                    wa2[j] = 0.0;
                    wa3[j] = diag[j] * wa1[j];
                }
                const double qnorm = enorm(n, wa3);
                if (qnorm > delta) {
                    int l = 0;
                    #pragma unroll
                    for (int j = 0; j < n; j++) {
                        const double temp = qtf[j];
                        #pragma unroll
                        for (int i = j; i < n; i++) {
                            wa2[i] += r[l] * temp;
                            l++;
                        }
                        wa2[j] = wa2[j] / diag[j];
                    }
                    const double gnorm = enorm(n, wa2);
                    double sgnorm = 0.0;
                    double alpha = delta / qnorm;
                    if (gnorm != 0.0) {
                        #pragma unroll
                        for (int j = 0; j < n; j++)
                            wa2[j] = (wa2[j] / gnorm) / diag[j];
                        l = 0;
                        #pragma unroll
                        for (int j = 0; j < n; j++) {
                            double sum = 0.0;
                            #pragma unroll
                            for (int i = j; i < n; i++) {
                                sum += r[l] * wa2[i];
                                l++;
                            }
                            wa3[j] = sum;
                        }
                        double temp = enorm(n, wa3);
                        sgnorm = (gnorm / temp) / temp;
                        alpha = 0.0;
                        if (sgnorm < delta) {
                            const double bnorm = enorm(n, qtf);
                            temp = (bnorm / gnorm) * (bnorm / qnorm) * (sgnorm / delta);
                            const double dq = delta / qnorm, sd = sgnorm / delta;
                            temp = temp - dq * (sd * sd) +
                                   sqrt((temp - dq) * (temp - dq) + (1.0 - dq * dq) * (1.0 - sd * sd));
                            alpha = (dq * (1.0 - sd * sd)) / temp;
                        }
                    }
                    const double temp = (1.0 - alpha) * (sgnorm < delta ? sgnorm : delta);
                    #pragma unroll
                    for (int j = 0; j < n; j++)
                        wa1[j] = temp * wa2[j] + alpha * wa1[j];
                }
            }
            #pragma unroll
            for (int j = 0; j < n; j++) {
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            const double pnorm = enorm(n, wa3);
            if (iter == 1)
                delta = delta < pnorm ? delta : pnorm;
            ok = fcn(wa2, wa4);
            nfev++;
            if (!ok)
                return -1;
            const double fnorm1 = enorm(n, wa4);
            double actred = -1.0;
            if (fnorm1 < fnorm) {
                const double q = fnorm1 / fnorm;
                actred = 1.0 - q * q;
            }
            {
                int l = 0;
                #pragma unroll
                for (int i = 0; i < n; i++) {
                    double sum = 0.0;
                    #pragma unroll
                    for (int j = i; j < n; j++) {
                        sum += r[l] * wa1[j];
                        l++;
                    }
                    wa3[i] = qtf[i] + sum;
                }
            }
            const double temp = enorm(n, wa3);
            double prered = 0.0;
            if (temp < fnorm) {
                const double q = temp / fnorm;
                prered = 1.0 - q * q;
            }
            double ratio = 0.0;
            if (prered > 0.0)
                ratio = actred / prered;
            if (ratio < 0.1) {
                ncsuc = 0;
                ncfail++;
                delta = 0.5 * delta;
            } else {
                ncfail = 0;
                ncsuc++;
                if (ratio >= 0.5 || ncsuc > 1)
                    delta = delta > pnorm / 0.5 ? delta : pnorm / 0.5;
                if (fabs(ratio - 1.0) <= 0.1)
                    delta = pnorm / 0.5;
            }
            if (ratio >= 1.0e-4) {
                #pragma unroll
                for (int j = 0; j < n; j++) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                    fvec[j] = wa4[j];
                }
                xnorm = enorm(n, wa2);
                fnorm = fnorm1;
                iter++;
            }
            nslow1++;
            if (actred >= 0.001)
                nslow1 = 0;
            if (jeval)
                nslow2++;
            if (actred >= 0.1)
                nslow2 = 0;
            if (delta <= xtol * xnorm || fnorm == 0.0)
                info = 1;
            if (info != 0)
                return info;
            if (nfev >= maxfev)
                info = 2;
            {
                const double a = 0.1 * delta > pnorm ? 0.1 * delta : pnorm;
                if (0.1 * a <= EPSMCH * xnorm)
                    info = 3;
            }
            if (nslow2 == 5)
                info = 4;
            if (nslow1 == 10)
                info = 5;
            if (info != 0)
                return info;
            if (ncfail == 2)
                break;
            // rank one modification of the jacobian
            #pragma unroll
            for (int j = 0; j < n; j++) {
                double sum = 0.0;
                #pragma unroll
                for (int i = 0; i < n; i++)
                    sum += FJ(i, j) * wa4[i];
                wa2[j] = (sum - wa3[j]) / pnorm;
                wa1[j] = diag[j] * ((diag[j] * wa1[j]) / pnorm);
                if (ratio >= 1.0e-4)
                    qtf[j] = sum;
            }
            // r1updt.f (m = n): u = wa1, v = wa2, w = wa3; 1-based bookkeeping as in the Fortran
            {
#define S(i) r[(i) - 1]
#define V(i) wa2[(i) - 1]
#define W(i) wa3[(i) - 1]
#define U(i) wa1[(i) - 1]
                int jj = (n * (n + 1)) / 2;
                W(n) = S(jj);
                #pragma unroll
                for (int nmj = 1; nmj <= n - 1; nmj++) {
                    const int j = n - nmj;
                    jj -= (n - j + 1);
                    W(j) = 0.0;
                    if (V(j) == 0.0)
                        continue;
                    double c, s, tau;
                    givens(V(n), V(j), c, s, tau);
                    V(n) = s * V(j) + c * V(n);
                    V(j) = tau;
                    int l = jj;
                    #pragma unroll
                    for (int i = j; i <= n; i++) {
                        const double temp = c * S(l) - s * W(i);
                        W(i) = s * S(l) + c * W(i);
                        S(l) = temp;
                        l++;
                    }
                }
                #pragma unroll
                for (int i = 1; i <= n; i++)
                    W(i) = W(i) + V(n) * U(i);
                #pragma unroll
                for (int j = 1; j <= n - 1; j++) {
                    if (W(j) != 0.0) {
                        double c, s, tau;
                        givens(S(jj), W(j), c, s, tau);
                        int l = jj;
                        #pragma unroll
                        for (int i = j; i <= n; i++) {
                            const double temp = c * S(l) + s * W(i);
                            W(i) = -s * S(l) + c * W(i);
                            S(l) = temp;
                            l++;
                        }
                        W(j) = tau;
                    }
                    jj += (n - j + 1);
                }
                S(jj) = W(n);
            }
#undef S
#undef V
#undef W
#undef U
            // r1mpyq.f on fjac (n x n) and on qtf (1 x n)
            #pragma unroll
            for (int nmj = 1; nmj <= n - 1; nmj++) {
                const int j = n - nmj - 1;
                double c, s;
                recover(wa2[j], c, s);
                #pragma unroll
                for (int i = 0; i < n; i++) {
                    const double temp = c * FJ(i, j) - s * FJ(i, n - 1);
                    FJ(i, n - 1) = s * FJ(i, j) + c * FJ(i, n - 1);
                    FJ(i, j) = temp;
                }
                const double temp = c * qtf[j] - s * qtf[n - 1];
                qtf[n - 1] = s * qtf[j] + c * qtf[n - 1];
                qtf[j] = temp;
            }
            #pragma unroll
            for (int j = 0; j < n - 1; j++) {
                double c, s;
                recover(wa3[j], c, s);
                #pragma unroll
                for (int i = 0; i < n; i++) {
                    const double temp = c * FJ(i, j) + s * FJ(i, n - 1);
                    FJ(i, n - 1) = -s * FJ(i, j) + c * FJ(i, n - 1);
                    FJ(i, j) = temp;
                }
                const double temp = c * qtf[j] + s * qtf[n - 1];
                qtf[n - 1] = -s * qtf[j] + c * qtf[n - 1];
                qtf[j] = temp;
            }
            jeval = false;
        }
    }
#undef FJ
}

}  // namespace hybrd
}  // namespace rox
