import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy
from pauxy_amd import systems, trial as tm, _lib as L
from pauxy_amd.qmc.afqmc import AFQMC
from pauxy_amd.context import release_context
s = systems.synthetic_generic(100, 500, (25, 25), seed=7)
t = tm.rhf_trial_generic(s)
orders = [int(x) for x in (sys.argv[1:] or ['6', '0'])]
for order in orders:
    opt = {'qmc': {'timestep': 0.005, 'num_steps': 10, 'blocks': 10**6, 'stabilise_freq': 10, 'pop_control_freq': 5,
                   'num_walkers': 256, 'rng_seed': 7},
           'propagator': {'device_rng': True, 'rng_seed': 7, 'expansion_order': order},
           'estimators': {'mixed': {'verbose': False}}}
    a = AFQMC(options=opt, system=s, trial=t)
    dev = a.psi.dev
    es = a.run_batched(10)
    dev.kernel_trace(True)
    a.run_batched(30, first_step=11, eshift=es)
    dev.sync()
    ms = dev.kernel_trace_get(L.K_PROPAGATOR)
    print(os.environ.get('AFQ_PF_DBG', '-'), order, len(ms), float(numpy.mean(ms)) * 1e3, 'us')
    dev.kernel_trace(False)
    release_context(s, t)
