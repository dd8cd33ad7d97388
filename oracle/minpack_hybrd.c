/* minpack_hybrd.c -- TEST INFRASTRUCTURE ONLY (part of the oracle library).
 *
 * The 2-D branch of the reference's chief-ray aiming calls
 * scipy.optimize.fsolve (rayoptics/raytr/trace.py:404-410), i.e. MINPACK's
 * `hybrd` (Powell's hybrid method: forward-difference Jacobian, QR, dog-leg
 * steps, Broyden rank-one updates of the factorisation).  MINPACK is a
 * third-party dependency of the reference (SciPy 1.15.3 in the build container,
 * scipy/optimize/_minpack) and is not part of /root/reference: this file
 * restates its published algorithm -- Argonne MINPACK-1, More / Garbow /
 * Hillstrom 1980: hybrd, fdjac1, qrfac, qform, dogleg, r1updt, r1mpyq, enorm,
 * dpmpar -- one C function per Fortran subroutine, same operations in the same
 * order, for any n <= HYBRD_MAXN.
 *
 * Parity status: PINNED against the installed scipy.optimize.fsolve
 * (tests/test_oracle_hybrd.py): x, fvec, info, nfev and the final factorisation
 * (fjac = Q, r, qtf) on random smooth and degenerate systems, and through the
 * reference's own iterate_ray on models with x fields (tests/test_dropin_reference.py).
 *
 * Compiled with -ffp-contract=off like the rest of the oracle.
 */
#include <math.h>
#include <string.h>

#include "minpack_hybrd.h"

static const double EPSMCH = 2.220446049250313e-16;     /* dpmpar(1) */
static const double GIANT = 1.7976931348623157e308;     /* dpmpar(3) */

/* enorm.f: Euclidean norm with the three-accumulator scaling */
static double enorm(int n, const double *x)
{
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    const double agiant = rgiant / (double)n;
    for (int i = 0; i < n; i++) {
        const double xabs = fabs(x[i]);
        if (xabs > rdwarf && xabs < agiant) {
            s2 += xabs * xabs;                          /* sum for intermediate components */
        } else if (xabs <= rdwarf) {                    /* sum for small components */
            if (xabs > x3max) {
                const double q = x3max / xabs;
                s3 = 1.0 + s3 * (q * q);
                x3max = xabs;
            } else if (xabs != 0.0) {
                const double q = xabs / x3max;
                s3 += q * q;
            }
        } else {                                        /* sum for large components */
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

/* column-major helpers: a(i,j), 0-based, leading dimension n */
#define A(a, i, j) (a)[(i) + (size_t)(j) * n]

/* fdjac1.f, dense case (ml + mu + 1 >= n): forward differences */
static int fdjac1(hybrd_fcn fcn, void *ctx, int n, double *x, const double *fvec, double *fjac,
                  double epsfcn, double *wa1)
{
    const double eps = sqrt(epsfcn > EPSMCH ? epsfcn : EPSMCH);
    for (int j = 0; j < n; j++) {
        const double temp = x[j];
        double h = eps * fabs(temp);
        if (h == 0.0)
            h = eps;
        x[j] = temp + h;
        const int iflag = fcn(n, x, wa1, ctx);
        if (iflag < 0)
            return iflag;
        x[j] = temp;
        for (int i = 0; i < n; i++)
            A(fjac, i, j) = (wa1[i] - fvec[i]) / h;
    }
    return 0;
}

/* qrfac.f with pivot = .false., m = n */
static void qrfac(int n, double *a, double *rdiag, double *acnorm)
{
    for (int j = 0; j < n; j++) {
        acnorm[j] = enorm(n, &A(a, 0, j));
        rdiag[j] = acnorm[j];
    }
    for (int j = 0; j < n; j++) {
        /* the householder transformation to reduce the j-th column to a multiple of e_j */
        double ajnorm = enorm(n - j, &A(a, j, j));
        if (ajnorm != 0.0) {
            if (A(a, j, j) < 0.0)
                ajnorm = -ajnorm;
            for (int i = j; i < n; i++)
                A(a, i, j) /= ajnorm;
            A(a, j, j) += 1.0;
            /* apply the transformation to the remaining columns */
            for (int k = j + 1; k < n; k++) {
                double sum = 0.0;
                for (int i = j; i < n; i++)
                    sum += A(a, i, j) * A(a, i, k);
                const double temp = sum / A(a, j, j);
                for (int i = j; i < n; i++)
                    A(a, i, k) -= temp * A(a, i, j);
            }
        }
        rdiag[j] = -ajnorm;
    }
}

/* qform.f, m = n: accumulate Q from its factored form */
static void qform(int n, double *q, double *wa)
{
    for (int j = 1; j < n; j++)
        for (int i = 0; i < j; i++)
            A(q, i, j) = 0.0;
    for (int l = 0; l < n; l++) {
        const int k = n - 1 - l;
        for (int i = k; i < n; i++) {
            wa[i] = A(q, i, k);
            A(q, i, k) = 0.0;
        }
        A(q, k, k) = 1.0;
        if (wa[k] == 0.0)
            continue;
        for (int j = k; j < n; j++) {
            double sum = 0.0;
            for (int i = k; i < n; i++)
                sum += A(q, i, j) * wa[i];
            const double temp = sum / wa[k];
            for (int i = k; i < n; i++)
                A(q, i, j) -= temp * wa[i];
        }
    }
}

/* dogleg.f: r is the upper triangle stored by rows */
static void dogleg(int n, const double *r, const double *diag, const double *qtb, double delta,
                   double *x, double *wa1, double *wa2)
{
    /* first, calculate the gauss-newton direction */
    int jj = (n * (n + 1)) / 2;                 /* (0-based index one past the last element) */
    for (int k = 1; k <= n; k++) {
        const int j = n - k;                    /* 0-based row */
        jj -= k;
        int l = jj + 1;
        double sum = 0.0;
        for (int i = j + 1; i < n; i++) {
            sum += r[l] * x[i];
            l++;
        }
        double temp = r[jj];
        if (temp == 0.0) {
            l = j;
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
        x[j] = (qtb[j] - sum) / temp;
    }
    /* test whether the gauss-newton direction is acceptable */
    for (int j = 0; j < n; j++) {
        wa1[j] = 0.0;
        wa2[j] = diag[j] * x[j];
    }
    const double qnorm = enorm(n, wa2);
    if (qnorm <= delta)
        return;
    /* the gauss-newton direction is not acceptable: the scaled gradient direction */
    int l = 0;
    for (int j = 0; j < n; j++) {
        const double temp = qtb[j];
        for (int i = j; i < n; i++) {
            wa1[i] += r[l] * temp;
            l++;
        }
        wa1[j] = wa1[j] / diag[j];
    }
    /* the norm of the scaled gradient; the special case in which it is zero */
    const double gnorm = enorm(n, wa1);
    double sgnorm = 0.0;
    double alpha = delta / qnorm;
    if (gnorm != 0.0) {
        /* the point along the scaled gradient at which the quadratic is minimized */
        for (int j = 0; j < n; j++)
            wa1[j] = (wa1[j] / gnorm) / diag[j];
        l = 0;
        for (int j = 0; j < n; j++) {
            double sum = 0.0;
            for (int i = j; i < n; i++) {
                sum += r[l] * wa1[i];
                l++;
            }
            wa2[j] = sum;
        }
        double temp = enorm(n, wa2);
        sgnorm = (gnorm / temp) / temp;
        /* test whether the scaled gradient direction is acceptable */
        alpha = 0.0;
        if (sgnorm < delta) {
            /* not acceptable: the point along the dogleg at which the quadratic is minimized */
            const double bnorm = enorm(n, qtb);
            temp = (bnorm / gnorm) * (bnorm / qnorm) * (sgnorm / delta);
            const double dq = delta / qnorm, sd = sgnorm / delta;
            temp = temp - dq * (sd * sd) +
                   sqrt((temp - dq) * (temp - dq) + (1.0 - dq * dq) * (1.0 - sd * sd));
            alpha = (dq * (1.0 - sd * sd)) / temp;
        }
    }
    /* convex combination of the gauss-newton and the scaled gradient directions */
    const double temp = (1.0 - alpha) * (sgnorm < delta ? sgnorm : delta);
    for (int j = 0; j < n; j++)
        x[j] = temp * wa1[j] + alpha * x[j];
}

/* r1updt.f, m = n: s (lower trapezoidal by columns = r by rows) + u v^T, back to triangular form
 * with Givens rotations whose parameters are left in v and w */
static void r1updt(int n, double *s, const double *u, double *v, double *w, int *sing)
{
    /* 1-based indices as in the Fortran */
#define S(i) s[(i) - 1]
#define V(i) v[(i) - 1]
#define W(i) w[(i) - 1]
#define U(i) u[(i) - 1]
    const int m = n;
    int jj = (n * (2 * m - n + 1)) / 2 - (m - n);
    /* move the nontrivial part of the last column of s into w */
    int l = jj;
    for (int i = n; i <= m; i++) {
        W(i) = S(l);
        l++;
    }
    /* rotate v into a multiple of the n-th unit vector, introducing a spike into w */
    for (int nmj = 1; nmj <= n - 1; nmj++) {
        const int j = n - nmj;
        jj -= (m - j + 1);
        W(j) = 0.0;
        if (V(j) == 0.0)
            continue;
        double c, sn, tau;
        if (fabs(V(n)) < fabs(V(j))) {
            const double cotan = V(n) / V(j);
            sn = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
            c = sn * cotan;
            tau = 1.0;
            if (fabs(c) * GIANT > 1.0)
                tau = 1.0 / c;
        } else {
            const double tn = V(j) / V(n);
            c = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
            sn = c * tn;
            tau = sn;
        }
        /* apply the transformation to v and keep what recovers the rotation */
        V(n) = sn * V(j) + c * V(n);
        V(j) = tau;
        /* apply the transformation to s and extend the spike in w */
        l = jj;
        for (int i = j; i <= m; i++) {
            const double temp = c * S(l) - sn * W(i);
            W(i) = sn * S(l) + c * W(i);
            S(l) = temp;
            l++;
        }
    }
    /* add the spike from the rank 1 update to w */
    for (int i = 1; i <= m; i++)
        W(i) = W(i) + V(n) * U(i);
    /* eliminate the spike */
    *sing = 0;
    for (int j = 1; j <= n - 1; j++) {
        if (W(j) != 0.0) {
            double c, sn, tau;
            if (fabs(S(jj)) < fabs(W(j))) {
                const double cotan = S(jj) / W(j);
                sn = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
                c = sn * cotan;
                tau = 1.0;
                if (fabs(c) * GIANT > 1.0)
                    tau = 1.0 / c;
            } else {
                const double tn = W(j) / S(jj);
                c = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
                sn = c * tn;
                tau = sn;
            }
            /* apply the transformation to s and reduce the spike in w */
            l = jj;
            for (int i = j; i <= m; i++) {
                const double temp = c * S(l) + sn * W(i);
                W(i) = -sn * S(l) + c * W(i);
                S(l) = temp;
                l++;
            }
            W(j) = tau;
        }
        if (S(jj) == 0.0)
            *sing = 1;
        jj += (m - j + 1);
    }
    /* move w back into the last column of the output s */
    l = jj;
    for (int i = n; i <= m; i++) {
        S(l) = W(i);
        l++;
    }
    if (S(jj) == 0.0)
        *sing = 1;
}
#undef S
#undef V
#undef W
#undef U

/* r1mpyq.f: a (m x n, column-major, leading dimension lda) times the 2(n-1) rotations of r1updt */
static void r1mpyq(int m, int n, double *a, int lda, const double *v, const double *w)
{
    for (int nmj = 1; nmj <= n - 1; nmj++) {
        const int j = n - nmj - 1;              /* 0-based */
        double c, sn;
        if (fabs(v[j]) > 1.0) {
            c = 1.0 / v[j];
            sn = sqrt(1.0 - c * c);
        } else {
            sn = v[j];
            c = sqrt(1.0 - sn * sn);
        }
        for (int i = 0; i < m; i++) {
            const double temp = c * a[i + (size_t)j * lda] - sn * a[i + (size_t)(n - 1) * lda];
            a[i + (size_t)(n - 1) * lda] = sn * a[i + (size_t)j * lda] + c * a[i + (size_t)(n - 1) * lda];
            a[i + (size_t)j * lda] = temp;
        }
    }
    for (int j = 0; j < n - 1; j++) {
        double c, sn;
        if (fabs(w[j]) > 1.0) {
            c = 1.0 / w[j];
            sn = sqrt(1.0 - c * c);
        } else {
            sn = w[j];
            c = sqrt(1.0 - sn * sn);
        }
        for (int i = 0; i < m; i++) {
            const double temp = c * a[i + (size_t)j * lda] + sn * a[i + (size_t)(n - 1) * lda];
            a[i + (size_t)(n - 1) * lda] = -sn * a[i + (size_t)j * lda] + c * a[i + (size_t)(n - 1) * lda];
            a[i + (size_t)j * lda] = temp;
        }
    }
}

/* hybrd.f with mode = 1 (internal scaling), ml = mu = n - 1 (dense Jacobian), nprint = 0 */
int rox_oracle_hybrd(hybrd_fcn fcn, void *ctx, int n, double *x, double *fvec, double xtol,
                     int maxfev, double epsfcn, double factor, int *nfev_out, double *fjac,
                     double *r, double *qtf)
{
    if (n <= 0 || n > HYBRD_MAXN || xtol < 0.0 || maxfev <= 0 || factor <= 0.0)
        return 0;
    double diag[HYBRD_MAXN], wa1[HYBRD_MAXN], wa2[HYBRD_MAXN], wa3[HYBRD_MAXN], wa4[HYBRD_MAXN];
    int info = 0, nfev;
    double xnorm = 0.0, delta = 0.0;
    /* evaluate the function at the starting point and calculate its norm */
    int iflag = fcn(n, x, fvec, ctx);
    nfev = 1;
    if (iflag < 0) {
        *nfev_out = nfev;
        return iflag;
    }
    double fnorm = enorm(n, fvec);
    const int msum = n;
    int iter = 1, ncsuc = 0, ncfail = 0, nslow1 = 0, nslow2 = 0;
    for (;;) {                                          /* outer loop */
        int jeval = 1;
        /* calculate the jacobian matrix */
        iflag = fdjac1(fcn, ctx, n, x, fvec, fjac, epsfcn, wa1);
        nfev += msum;
        if (iflag < 0)
            break;
        /* compute the qr factorization of the jacobian */
        qrfac(n, fjac, wa1, wa2);
        if (iter == 1) {
            /* scale according to the norms of the columns of the initial jacobian */
            for (int j = 0; j < n; j++) {
                diag[j] = wa2[j];
                if (wa2[j] == 0.0)
                    diag[j] = 1.0;
            }
            /* the norm of the scaled x; initialize the step bound delta */
            for (int j = 0; j < n; j++)
                wa3[j] = diag[j] * x[j];
            xnorm = enorm(n, wa3);
            delta = factor * xnorm;
            if (delta == 0.0)
                delta = factor;
        }
        /* form (q transpose)*fvec and store in qtf */
        for (int i = 0; i < n; i++)
            qtf[i] = fvec[i];
        for (int j = 0; j < n; j++) {
            if (A(fjac, j, j) != 0.0) {
                double sum = 0.0;
                for (int i = j; i < n; i++)
                    sum += A(fjac, i, j) * qtf[i];
                const double temp = -sum / A(fjac, j, j);
                for (int i = j; i < n; i++)
                    qtf[i] += A(fjac, i, j) * temp;
            }
        }
        /* copy the triangular factor of the qr factorization into r */
        int sing = 0;
        for (int j = 0; j < n; j++) {
            int l = j;
            for (int i = 0; i < j; i++) {
                r[l] = A(fjac, i, j);
                l += n - 1 - i;
            }
            r[l] = wa1[j];
            if (wa1[j] == 0.0)
                sing = 1;
        }
        (void)sing;
        /* accumulate the orthogonal factor in fjac */
        qform(n, fjac, wa1);
        /* rescale if necessary */
        for (int j = 0; j < n; j++)
            diag[j] = diag[j] > wa2[j] ? diag[j] : wa2[j];
        for (;;) {                                      /* inner loop */
            /* determine the direction p */
            dogleg(n, r, diag, qtf, delta, wa1, wa2, wa3);
            /* store the direction p and x + p; the norm of p */
            for (int j = 0; j < n; j++) {
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            const double pnorm = enorm(n, wa3);
            /* on the first iteration, adjust the initial step bound */
            if (iter == 1)
                delta = delta < pnorm ? delta : pnorm;
            /* evaluate the function at x + p and calculate its norm */
            iflag = fcn(n, wa2, wa4, ctx);
            nfev++;
            if (iflag < 0)
                goto done;
            const double fnorm1 = enorm(n, wa4);
            /* the scaled actual reduction */
            double actred = -1.0;
            if (fnorm1 < fnorm) {
                const double q = fnorm1 / fnorm;
                actred = 1.0 - q * q;
            }
            /* the scaled predicted reduction */
            int l = 0;
            for (int i = 0; i < n; i++) {
                double sum = 0.0;
                for (int j = i; j < n; j++) {
                    sum += r[l] * wa1[j];
                    l++;
                }
                wa3[i] = qtf[i] + sum;
            }
            const double temp = enorm(n, wa3);
            double prered = 0.0;
            if (temp < fnorm) {
                const double q = temp / fnorm;
                prered = 1.0 - q * q;
            }
            /* the ratio of the actual to the predicted reduction */
            double ratio = 0.0;
            if (prered > 0.0)
                ratio = actred / prered;
            /* update the step bound */
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
            /* test for successful iteration */
            if (ratio >= 1.0e-4) {
                /* successful iteration: update x, fvec, and their norms */
                for (int j = 0; j < n; j++) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                    fvec[j] = wa4[j];
                }
                xnorm = enorm(n, wa2);
                fnorm = fnorm1;
                iter++;
            }
            /* determine the progress of the iteration */
            nslow1++;
            if (actred >= 0.001)
                nslow1 = 0;
            if (jeval)
                nslow2++;
            if (actred >= 0.1)
                nslow2 = 0;
            /* test for convergence */
            if (delta <= xtol * xnorm || fnorm == 0.0)
                info = 1;
            if (info != 0)
                goto done;
            /* tests for termination and stringent tolerances */
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
                goto done;
            /* criterion for recalculating the jacobian approximation by forward differences */
            if (ncfail == 2)
                break;
            /* the rank one modification to the jacobian; update qtf if necessary */
            for (int j = 0; j < n; j++) {
                double sum = 0.0;
                for (int i = 0; i < n; i++)
                    sum += A(fjac, i, j) * wa4[i];
                wa2[j] = (sum - wa3[j]) / pnorm;
                wa1[j] = diag[j] * ((diag[j] * wa1[j]) / pnorm);
                if (ratio >= 1.0e-4)
                    qtf[j] = sum;
            }
            /* the qr factorization of the updated jacobian */
            r1updt(n, r, wa1, wa2, wa3, &sing);
            r1mpyq(n, n, fjac, n, wa2, wa3);
            r1mpyq(1, n, qtf, 1, wa2, wa3);
            jeval = 0;
        }
    }
done:
    if (iflag < 0)
        info = iflag;
    *nfev_out = nfev;
    return info;
}
