// Weight update of one walker at the end of a step (propagation/continuous.py:264-292 hybrid, :294-318 local energy,
// :194-200 free projection), the driver's weight cap and the estimator terms that ride on it.  Device code shared by the
// kernels that finish a step's overlap: greens_small_kernel / weight_kernel (k_small.hip) and det_combine_kernel
// (k_bigdet.hip).
#pragma once
#include "afq_internal.h"

__device__ inline cplx clog_(cplx z) { return cmake(log(hypot(z.x, z.y)), atan2(z.y, z.x)); }
__device__ inline cplx cexp_(cplx z) {
    const double e = exp(z.x);
    double s, c; sincos(z.y, &s, &c);
    return cmake(e * c, e * s);
}

struct WeightArgs {
    int nw, flags;
    double dt;
    cplx eshift;
    const int *alive;
    const cplx *ovlp_old, *ovlp_new, *cmf, *cfb;
    double *weight;
    cplx *ot, *ehyb, *phase, *eloc;
    const cplx *energy;         // [nw, 3] local energy of the walker before the step (hybrid == false)
    unsigned long long *counters;
    // back-propagation bookkeeping (continuous.py:284-289,310-315, walkers/stack.py:51-76); null = off
    int *bp_flag;
    double *bp_cos;
    cplx *bp_ph;
    // weight cap of the driver (qmc/afqmc.py:235-236) applied right behind the update; cap_frac <= 0: off
    double cap_frac, cap_total;
    const double *cap_total_dev;    // total weight of the last comb when cap_total < 0
    // use_log_shift (walkers/single_det.py:192): walker.ot = overlap * exp(-log_shift); the ratios of a step do not
    // see it, both of its overlaps carry the same shift.  1 when the option is off.
    double ot_scale;
    // afq_estimates_fuse_next: per-walker accumulators [nw][6] of the estimator terms of this step (null = off)
    double *est_acc;
    const double *unscaled;
};

__device__ inline void bp_record(const WeightArgs &a, int w, double magn, cplx wfac0, double cosine_fac) {
    if (!a.bp_flag) return;
    if (!(magn > 1e-16)) { wfac0 = cmake(0.0, 0.0); cosine_fac = 0.0; }
    a.bp_flag[w] = 1;
    a.bp_cos[w] *= cosine_fac;
    a.bp_ph[w] = cmul(a.bp_ph[w], wfac0);
}

// propagation/continuous.py:264-292 (hybrid) and :194-200 (free projection)
__device__ static void weight_update(const WeightArgs &a, const int w) {
    if (a.bp_flag) a.bp_flag[w] = 0;
    if (!a.alive[w]) return;
    const cplx on = a.ovlp_new[w];
    if (a.flags & AFQ_PROP_FREE_PROJECTION) {
        const cplx e = cexp_(cmake(a.cmf[w].x + a.dt * a.eshift.x, a.cmf[w].y + a.dt * a.eshift.y));
        const double magn = hypot(e.x, e.y), dth = atan2(e.y, e.x);
        a.weight[w] *= magn;
        double s, c; sincos(dth, &s, &c);
        a.phase[w] = cmul(a.phase[w], cmake(c, s));
        a.ot[w] = cscale(on, a.ot_scale);
        return;
    }
    const cplx ratio = cdiv(on, a.ovlp_old[w]);
    if (!(a.flags & AFQ_PROP_HYBRID)) {
        // local-energy weight update, propagation/continuous.py:294-318 (+ :216-230)
        const cplx el = a.energy[3 * w];
        double re = el.x;
        const double ebound = sqrt(2.0 / a.dt);
        if (hypot(a.eshift.x, a.eshift.y) >= 1e-10) {
            if (re > a.eshift.x + ebound) { re = a.eshift.x + ebound; atomicAdd(&a.counters[1], 1ull); }
            else if (re < a.eshift.x - ebound) { re = a.eshift.x - ebound; atomicAdd(&a.counters[1], 1ull); }
        }
        const double magn = exp(-0.5 * a.dt * (re + a.eloc[w].x - a.eshift.x));
        const double wfac_imag = exp(-0.5 * a.dt * (el.y + a.eloc[w].y - a.eshift.y));   // continuous.py:299
        a.eloc[w] = el;
        a.ot[w] = cscale(on, a.ot_scale);
        if (!isinf(magn)) {
            const double cf = fmax(0.0, cos(atan2(ratio.y, ratio.x)));
            a.weight[w] *= magn * cf;
            bp_record(a, w, magn, cmake(wfac_imag, 0.0), cf);
        } else a.weight[w] = 0.0;
        return;
    }
    const cplx lg = clog_(ratio);
    cplx eh = cmake(-(lg.x + a.cfb[w].x + a.cmf[w].x) / a.dt, -(lg.y + a.cfb[w].y + a.cmf[w].y) / a.dt);
    const double ebound = sqrt(2.0 / a.dt);
    if (hypot(a.eshift.x, a.eshift.y) >= 1e-10) {       // continuous.py:206
        if (eh.x > a.eshift.x + ebound) { eh.x = a.eshift.x + ebound; atomicAdd(&a.counters[1], 1ull); }
        else if (eh.x < a.eshift.x - ebound) { eh.x = a.eshift.x - ebound; atomicAdd(&a.counters[1], 1ull); }
    }
    const cplx old = a.ehyb[w];
    const cplx arg = cmake(-a.dt * (0.5 * (eh.x + old.x) - a.eshift.x), -a.dt * (0.5 * (eh.y + old.y) - a.eshift.y));
    const cplx imp = cexp_(arg);
    const double magn = hypot(imp.x, imp.y);
    a.ehyb[w] = eh;
    a.ot[w] = cscale(on, a.ot_scale);
    if (!isinf(magn)) {
        const double dtheta = -a.dt * eh.y - a.cfb[w].y;
        const double cf = fmax(0.0, cos(dtheta));
        a.weight[w] *= magn * cf;
        bp_record(a, w, magn, cmake(imp.x / magn, imp.y / magn), cf);
    } else {
        a.weight[w] = 0.0;
    }
}

__device__ __attribute__((always_inline)) static void weight_update_and_cap(const WeightArgs &a, const int w) {
    weight_update(a, w);
    if (a.cap_frac > 0.0) {
        // every walker, propagated or not, exactly like the driver's loop
        const double cap = a.cap_frac * (a.cap_total < 0.0 ? a.cap_total_dev[0] : a.cap_total);
        if (fabs(a.weight[w]) > cap) a.weight[w] = cap;
    }
    if (a.est_acc) {
        // the terms estimates_kernel would add for this walker right behind this update (estimators/mixed.py:151-175,
        // 211-225, without the energy), summed over the STEPS of this walker slot here and over the walkers later
        const double x = a.weight[w];
        const cplx o = a.ot[w];
        cplx wf = cmake(x, 0.0);
        if (a.flags & AFQ_PROP_FREE_PROJECTION) wf = cscale(cmul(o, a.phase[w]), x);
        const cplx eh = cmul(wf, a.ehyb[w]);
        double *acc = a.est_acc + 6 * (long)w;
        acc[0] += a.unscaled[w];
        acc[1] += wf.x; acc[2] += wf.y;
        acc[3] += x * hypot(o.x, o.y);
        acc[4] += eh.x; acc[5] += eh.y;
    }
}

