/* minpack_hybrd.h -- TEST INFRASTRUCTURE ONLY: MINPACK hybrd restated (minpack_hybrd.c). */
#ifndef ROX_ORACLE_MINPACK_HYBRD_H
#define ROX_ORACLE_MINPACK_HYBRD_H

#define HYBRD_MAXN 8

/* fvec = f(x); a negative return value stops the iteration (MINPACK's iflag < 0) */
typedef int (*hybrd_fcn)(int n, const double *x, double *fvec, void *ctx);

/* hybrd with mode = 1, ml = mu = n - 1, nprint = 0 -- what scipy.optimize.fsolve(func, x0,
 * xtol=, maxfev=, epsfcn=, factor=) runs after its own extra evaluation of func(x0).
 * fjac [n*n, column-major] = Q, r [n(n+1)/2] = upper triangle by rows, qtf [n] as MINPACK
 * leaves them.  Returns info (a negative value = the callback's). */
int rox_oracle_hybrd(hybrd_fcn fcn, void *ctx, int n, double *x, double *fvec, double xtol,
                     int maxfev, double epsfcn, double factor, int *nfev, double *fjac,
                     double *r, double *qtf);

#endif
