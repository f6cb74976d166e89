"""Stage-by-stage comparison of AFQMC.run and AFQMC.run_batched at the C4 size around step 10 (debug helper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pauxy_amd import _lib as L
from pauxy_amd.context import release_context
from tests.test_gpu_batched import c4_afqmc


def go(batched):
    afqmc, s, t = c4_afqmc()
    psi, dev = afqmc.psi, afqmc.psi.dev
    nw = psi.nw
    afqmc.ride_estimates = False
    numpy.random.seed(1234)
    dev.set(L.F_WEIGHT, numpy.exp(0.6 * numpy.random.RandomState(5).normal(size=nw)))
    psi._invalidate()
    rec, cur = {}, {'step': 0, 'n': 0}

    def snap(tag):
        if cur['step'] in (9, 10):
            k = '%02d.%02d.%s' % (cur['step'], cur['n'], tag)
            cur['n'] += 1
            rec[k + '.ghalf'] = dev.get(L.F_GHALF)
            rec[k + '.phi'] = dev.get(L.F_PHI)

    def wrap(obj, name, tag):
        f = getattr(obj, name)

        def g(*a, **k):
            r = f(*a, **k)
            snap(tag)
            return r
        setattr(obj, name, g)

    calls = rec.setdefault('calls', [])
    for name in dir(dev):
        f = getattr(dev, name)
        if name.startswith('_') or not callable(f):
            continue

        def mk(name, f):
            def g(*a, **k):
                if cur['step'] < 12:
                    calls.append('%d:%s%s' % (cur['step'], name, tuple(x for x in a if isinstance(x, (int, bool, float)))))
                return f(*a, **k)
            return g
        setattr(dev, name, mk(name, f))

    P = afqmc.propagators
    for nm in ('_propagate_walker', 'propagate_walkers'):
        def mk2(nm, f):
            def g(*a, **k):
                if cur['step'] < 12:
                    calls.append('%d:P.%s' % (cur['step'], nm))
                return f(*a, **k)
            return g
        setattr(P, nm, mk2(nm, getattr(P, nm)))

    if batched:
        wrap(dev, 'reortho', 'reortho')
        wrap(afqmc.propagators, 'propagate_walkers', 'prop')
        wrap(psi, 'pop_control', 'comb')
        wrap(dev, 'estimates_update', 'est')
    else:
        wrap(psi, 'orthogonalise', 'reortho')
        wrap(psi, 'pop_control', 'comb')
        wrap(afqmc.estimators, 'update', 'est')
        w_last = psi.walkers[-1]
        pw = afqmc.propagators.propagate_walker

        def pw2(w, *a, **k):
            if w._i < 2:
                calls.append('%d:pw2(%d,pending=%s)' % (cur['step'], w._i, w._pending))
            r = pw(w, *a, **k)
            if w is w_last:
                snap('prop')
            return r
        afqmc.propagators.propagate_walker = pw2
        print("run: nwalkers in list", len(psi.walkers), "first weights", [psi.walkers[i].weight for i in range(3)])
        print("run: propagators", type(afqmc.propagators).__name__, "same dev", afqmc.propagators.dev is dev,
              "run uses", afqmc.propagators.propagate_walker is pw2)

    def on_step(step, psi_):
        cur['step'] = step + 1
        cur['n'] = 0

    cur['step'] = 1
    if batched:
        afqmc.run_batched(on_step=on_step, fetch_popcontrol=True)
    else:
        afqmc.run(on_step=on_step)
    release_context(s, t)
    return rec


a = go(False)
b = go(True)
print("run calls    :", ' '.join(a.pop('calls')))
print("batched calls:", ' '.join(b.pop('calls')))
print("run stages    :", sorted(set(k.rsplit('.', 1)[0] for k in a)))
print("batched stages:", sorted(set(k.rsplit('.', 1)[0] for k in b)))
ta = {}
for k in sorted(a):
    st, n, tag, f = k.split('.')
    ta.setdefault((st, tag, f), []).append(k)
tb = {}
for k in sorted(b):
    st, n, tag, f = k.split('.')
    tb.setdefault((st, tag, f), []).append(k)
for key in sorted(ta):
    if key not in tb:
        continue
    x, y = a[ta[key][-1]], b[tb[key][-1]]
    d = numpy.abs(x - y)
    print(key, "equal" if not d.any() else "max abs diff %.3e (max |x| %.3e), walkers differing %d" % (
        d.max(), numpy.abs(x).max(), int((d.reshape(d.shape[0], -1).max(axis=1) > 0).sum())))
