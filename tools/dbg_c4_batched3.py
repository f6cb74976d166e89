"""First step at which AFQMC.run and AFQMC.run_batched part on the 16 x 16 lattice with every walker alive (debug helper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from tests.test_gpu_batched import run_c3, c4_small_step_afqmc

a, ba, pa = run_c3(False, True, make=c4_small_step_afqmc)
b, bb, pb = run_c3(True, True, make=c4_small_step_afqmc)
for key in ('weight', 'ot', 'ehyb'):
    for st in range(a[key].shape[0]):
        d = numpy.abs(a[key][st] - b[key][st])
        if d.any():
            print(key, "first differs after step", st + 1, "max abs diff %.3e" % d.max(), "rel %.3e" % (d.max() / numpy.abs(a[key][st]).max()),
                  "walkers", int((d > 0).sum()), "min |weight|", numpy.abs(a['weight'][st]).min())
            break
    else:
        print(key, "equal throughout")
print("pix equal", numpy.array_equal(a['pix'], b['pix']))
print("weights min per step", numpy.abs(a['weight']).min(axis=1))
