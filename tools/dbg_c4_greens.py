"""Green's function / overlap / force bias of the N = 128 path against the oracle for several populations (debug helper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from oracle import afqmc_ref as ref
from pauxy_amd import _lib as L
from tests.helpers import make_device
from tests.test_gpu_fullsize import hubbard_c4

model = hubbard_c4()
M, na, nb = model.M, model.na, model.nb
for nw in (32, 64, 96, 128, 256):
    rng = numpy.random.RandomState(5)
    phis = model.psi[None] + 0.05 * (rng.rand(nw, M, na + nb) + 1j * rng.rand(nw, M, na + nb))
    dev = make_device(model, nw)
    dev.set(L.F_PHI, phis)
    ov = dev.calc_overlap()
    det = dev.greens(want_G=False)
    gh = dev.get(L.F_GHALF)
    xbar = dev.force_bias()
    worst = [0, 0, 0, 0]
    bad = []
    for w in sorted(set([0, 1, nw // 2, nw - 1] + list(range(0, nw, max(1, nw // 16))))):
        d, ghr, Gr = ref.greens_function(phis[w], model.psi, na, nb)
        e = [abs(ov[w] - d) / abs(d), abs(det[w] - d) / abs(d), numpy.abs(gh[w] - numpy.concatenate([ghr[0], ghr[1]]) if isinstance(ghr, (list, tuple)) else gh[w] - ghr).max(),
             numpy.abs(xbar[w] - model.force_bias(ghr, Gr)).max()]
        worst = [max(a, b) for a, b in zip(worst, e)]
        if max(e) > 1e-8:
            bad.append((w, ["%.1e" % x for x in e]))
    print("nw", nw, "worst rel/abs errors: overlap %.2e  det %.2e  ghalf %.2e  xbar %.2e" % tuple(worst), "bad walkers:", bad[:6])
    dev.close()
