/* spot_diagram.c -- libroxtrace.so from plain C: no Python, no torch, no HIP headers.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/spot_diagram.c -Lray-optics_amd -lroxtrace \
 *       -Wl,-rpath,$PWD/ray-optics_amd -lm -o /tmp/spot_diagram
 *   /tmp/spot_diagram [num [pairs.bin]]
 *
 * A biconvex N-BK7 singlet at finite conjugates, typed in as the table
 * rayoptics.seq.sequential.SequentialModel.path() would yield it (one row per interface,
 * object and image included; row i carries the gap and transform from interface i to i+1),
 * and the spot diagram of two object points: what
 *
 *   sm.trace_grid(spot, fi, wl, num_rays=num, form='list', append_if_none=False)
 *
 * returns in the reference (rayoptics/seq/sequential.py:1058-1085 with the `spot` filter of
 * rayoptics/mpl/axisarrayfigure.py:229-238) -- the (x, y) of every ray that reaches the image,
 * packed in ray order.  The kernel writes the pairs straight into this process's (page-locked)
 * host memory; rox_synchronize waits for it.  tests/test_c_example.py compiles this file and, on
 * a GPU, compares the pairs it dumps with the CPU oracle's for the same table, bit for bit. */
#define _POSIX_C_SOURCE 200112L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "roxtrace.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != 0) {                                                          \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rox_last_error());     \
            return 1;                                                            \
        }                                                                        \
    } while (0)

enum { N_IFCS = 4 };

/* the prescription: curvature, distance to the next interface, index behind the interface */
static const double CV[N_IFCS] = { 0.0, 1.0 / 50.0, -1.0 / 50.0, 0.0 };
static const double THI[N_IFCS] = { 100.0, 5.0, 95.0, 0.0 };
static const double NDX[N_IFCS] = { 1.0, 1.5168, 1.0, 1.0 };
static const double SEMI_AP[N_IFCS] = { 1.0e10, 9.0, 9.0, 1.0e10 };

static void make_rows(rox_surface rows[N_IFCS])
{
    memset(rows, 0, sizeof(rox_surface) * N_IFCS);
    for (int i = 0; i < N_IFCS; ++i) {
        rox_surface *s = &rows[i];
        s->mode = (i == 0 || i == N_IFCS - 1) ? ROX_DUMMY : ROX_TRANSMIT;
        s->profile = ROX_SPHERICAL;
        s->rt_order = ROX_RT_C_ORDER;           /* np.identity(3) */
        s->cv = CV[i];
        s->ec = 1.0;                            /* cc + 1.0 */
        s->rt[0] = s->rt[4] = s->rt[8] = 1.0;
        s->t[2] = THI[i];
        s->z_dir = 1.0;
        s->max_aperture = SEMI_AP[i];
        s->ph.kind = ROX_PH_NONE;
    }
}

int main(int argc, char **argv)
{
    const int num = argc > 1 ? atoi(argv[1]) : 256;
    const char *dump = argc > 2 ? argv[2] : NULL;
    if (num < 2 || num > 4096) {
        fprintf(stderr, "num must be in 2..4096\n");
        return 2;
    }
    if (rox_abi_version() != ROX_ABI_VERSION) {
        fprintf(stderr, "libroxtrace.so has ABI %d, this file was written against %d\n",
                rox_abi_version(), ROX_ABI_VERSION);
        return 2;
    }
    int n_dev = 0;
    CHECK(rox_device_count(&n_dev));
    if (n_dev < 1) {
        fprintf(stderr, "no MI355X visible\n");
        return 3;
    }
    CHECK(rox_set_device(0));

    rox_surface rows[N_IFCS];
    make_rows(rows);
    const double wvls[1] = { 587.5618 };
    rox_system *sys = NULL;
    CHECK(rox_system_create(rows, N_IFCS, NDX, wvls, 1, &sys));

    /* the pairs and their count land in this process's memory: page-align, page-lock */
    const size_t n_rays = (size_t)num * num;
    const size_t bytes = (n_rays * 2 * sizeof(double) + 4095) & ~(size_t)4095;
    double *pairs = NULL;
    int64_t *count = NULL;
    if (posix_memalign((void **)&pairs, 4096, bytes) || posix_memalign((void **)&count, 4096, 4096))
        return 4;
    void *d_pairs = NULL, *d_count = NULL;
    CHECK(rox_pin_host_memory(pairs, bytes, &d_pairs));
    CHECK(rox_pin_host_memory(count, 4096, &d_count));

    FILE *f = dump ? fopen(dump, "wb") : NULL;
    const double object_y[2] = { 0.0, 5.0 };
    for (int fi = 0; fi < 2; ++fi) {
        /* ray_start_from_osp's constants for this object point (opticalspec.py:358-366): the
         * stop is at the first surface, so the paraxial entrance pupil sits on it */
        rox_field fld;
        memset(&fld, 0, sizeof fld);
        fld.kind = ROX_FLD_EPD;
        fld.pt0[1] = object_y[fi];
        fld.eprad = 8.0;
        fld.z_enp = THI[0];
        fld.z_dir0 = 1.0;

        rox_grid grid = { { -1.0, -1.0 }, { 1.0, 1.0 }, num, ROX_GRID_PRODUCT, 0, 0 };
        rox_opts opts;
        memset(&opts, 0, sizeof opts);
        opts.flags = ROX_CHECK_APERTURES | ROX_INTERSECT_OBJ | ROX_APPLY_VIGNETTING;
        opts.out_mode = ROX_OUT_HITS_COMPACT;
        opts.first_surf = 1;
        opts.last_surf = N_IFCS - 2;
        opts.eps = 1.0e-12;
        opts.fuzz = 1.0e-5;

        rox_out out;
        memset(&out, 0, sizeof out);
        out.seg = (double *)d_pairs;
        out.n_hits = (int64_t *)d_count;
        out.ld = (int64_t)n_rays;               /* room for every ray */

        CHECK(rox_trace_pupil_grid(sys, &fld, &grid, 0, &opts, &out, NULL));
        CHECK(rox_synchronize(NULL));

        const int64_t n = *count;
        double cx = 0, cy = 0;
        for (int64_t k = 0; k < n; ++k) {
            cx += pairs[2 * k];
            cy += pairs[2 * k + 1];
        }
        cx /= (double)(n ? n : 1);
        cy /= (double)(n ? n : 1);
        double r2 = 0;
        for (int64_t k = 0; k < n; ++k)
            r2 += (pairs[2 * k] - cx) * (pairs[2 * k] - cx) + (pairs[2 * k + 1] - cy) * (pairs[2 * k + 1] - cy);
        printf("object y = %4.1f mm: %lld of %zu rays reach the image, centroid (%.6f, %.6f) mm, "
               "rms spot radius %.6f mm\n", object_y[fi], (long long)n, n_rays, cx, cy,
               sqrt(r2 / (double)(n ? n : 1)));
        if (f) {
            fwrite(&n, sizeof n, 1, f);
            fwrite(pairs, 2 * sizeof(double), (size_t)n, f);
        }
    }
    if (f)
        fclose(f);
    CHECK(rox_unpin_host_memory(pairs));
    CHECK(rox_unpin_host_memory(count));
    CHECK(rox_system_destroy(sys));
    free(pairs);
    free(count);
    return 0;
}
