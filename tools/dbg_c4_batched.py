"""Where do AFQMC.run and AFQMC.run_batched part at the C4 size?  (debug helper)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from tests.test_gpu_batched import run_c3, c4_afqmc

a, ba, pa = run_c3(False, True, make=c4_afqmc)
b, bb, pb = run_c3(True, True, make=c4_afqmc)
b2, bb2, pb2 = run_c3(True, True, make=c4_afqmc)
a2, ba2, pa2 = run_c3(False, True, make=c4_afqmc)
for name, x, y in (("run vs batched", a, b), ("batched vs batched", b, b2), ("run vs run", a, a2)):
    for key in ('weight', 'ot', 'ehyb'):
        d = x[key] != y[key]
        if d.any():
            st, w = numpy.argwhere(d)[0]
            rel = abs(x[key][st, w] - y[key][st, w]) / abs(x[key][st, w])
            print(name, key, "first diff at step", st + 1, "walker", w, "rel", rel, "count", d.sum(), "steps with diffs", sorted(set(numpy.argwhere(d)[:, 0] + 1)))
        else:
            print(name, key, "bit-equal")
    print(name, "pix equal", numpy.array_equal(x['pix'], y['pix']))
