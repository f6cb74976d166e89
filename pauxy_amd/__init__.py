"""pauxy_amd: MI355X-native phaseless-AFQMC walker propagation behind PAUXY's
Propagator / Walkers / Estimators surface.  The arithmetic lives in
``libafqmc_hip.so`` (hand-written HIP for gfx950); see DESIGN.md."""
__version__ = "0.1.0"
