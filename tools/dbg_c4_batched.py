"""Where do AFQMC.run and AFQMC.run_batched part at the C4 size?  (debug helper)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pauxy_amd import _lib as L
from pauxy_amd.context import release_context
from tests.test_gpu_batched import c4_afqmc


def go(batched, force_greens=False):
    afqmc, s, t = c4_afqmc()
    nw = afqmc.psi.nw
    numpy.random.seed(1234)
    afqmc.psi.dev.set(L.F_WEIGHT, numpy.exp(0.6 * numpy.random.RandomState(5).normal(size=nw)))
    afqmc.psi._invalidate()
    rec = {}

    def on_step(step, psi):
        if step in (9, 10, 11):
            rec['ghalf%d' % step] = psi.dev.get(L.F_GHALF)
            rec['phi%d' % step] = psi.dev.get(L.F_PHI)
            rec['ot%d' % step] = psi.dev.get(L.F_OT)
            rec['w%d' % step] = psi.dev.get(L.F_WEIGHT)
        if force_greens:
            psi.dev.greens(want_G=False)

    if batched:
        afqmc.run_batched(on_step=on_step, fetch_popcontrol=True)
    else:
        afqmc.run(on_step=on_step)
    release_context(s, t)
    return rec


a = go(False)
b = go(True)
c = go(True, force_greens=True)
for name, x, y in (("run vs batched", a, b), ("run vs batched+forced greens", a, c), ("batched vs batched+forced", b, c)):
    for k in sorted(x):
        d = numpy.abs(x[k] - y[k])
        print(name, k, "equal" if not d.any() else "max abs diff %.3e (max |x| %.3e), walkers differing %d" % (
            d.max(), numpy.abs(x[k]).max(), int((d.reshape(d.shape[0], -1).max(axis=1) > 0).sum())))
print("weights step 10:", numpy.sort(a['w10'])[:5], numpy.sort(a['w10'])[-5:], "alive", int((numpy.abs(a['w10']) > 1e-8).sum()))
