"""prop_ueg_kernel time against the Taylor order (fixed cost vs cost per product); debug helper: run under rocprofv3 --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pauxy_amd import systems, trial as trial_mod
from tools.bench_configs import run
order = int(sys.argv[1])
s = systems.UEG(2.0, 7, 7, 4.0)
run("C2 order %d" % order, s, trial_mod.hartree_fock_ueg(s), 256, 0.005, 40, 10, prop={'expansion_order': order})
