"""TEST INFRASTRUCTURE ONLY: a numpy stand-in for ``pauxy_amd.device.AfqDevice`` on top of ``oracle/afqmc_ref.py``.

Why it exists.  The plug-in classes of ``pauxy_amd`` (``Continuous``, ``Walkers``, ``Estimators`` ...) are host
logic -- mirrors, lazy batched launches, the once-per-sweep propagation trick, file output -- over an object with
the ``AfqDevice`` method surface.  With this stand-in that host logic can be driven where there is no GPU:

* in the build container by the GENUINE ``pauxy.qmc.afqmc.AFQMC`` (``tests/golden/make_golden.py dropin``: the
  driver's three plug-in imports switched as INTEGRATION.md section 3 says, nothing else), which proves the
  drop-in boundary against the reference's own driver code and records what that driver touches;
* in the travelling CPU suite by the restated loop (``tests/test_dropin_cpu.py``) against the golden trajectories.

It is installed by rebinding ``pauxy_amd.context.AfqDevice`` from a test; nothing under ``pauxy_amd/`` imports
it (or the oracle), and the ``-m gpu`` tests never use it: they run the same classes over the HIP library.

Coverage: single-determinant walkers, continuous Hubbard-Stratonovich fields (Generic, Hubbard charge / spin, UEG),
phaseless hybrid / local-energy weights, free projection, comb, mixed estimator, back-propagated one-body RDM, weight cap,
log shifts.  Per-walker
arithmetic is the oracle's, one walker at a time, in walker order.
"""
import numpy

from oracle import afqmc_ref as ref
from pauxy_amd import _lib as L


class OracleDevice(object):
    buffer_device = 'cpu'
    instances = []                    # every stand-in built (tests look at the call log of the last one)

    def __init__(self, device_id=0):
        self.device_id = device_id
        self.kind = None
        self.nw = 0
        self.ndet = 1
        self.nv = 1
        self.calls = []               # (method name) of every "launch": what a test asserts batching on
        self._sys = {}
        self._model = None
        self._w = []
        self._est = numpy.zeros(10, dtype=numpy.complex128)
        self._ntrig = numpy.zeros(4, dtype=numpy.int64)
        self._cap = 0.0
        self._cap_total = -1.0
        self._log_shift = (False, 0.0, 0.0)
        self._free = False
        self._hybrid = True
        OracleDevice.instances.append(self)

    # -- plumbing ---------------------------------------------------------
    def close(self):
        self._w = []

    def sync(self):
        pass

    def _log(self, name):
        self.calls.append(name)

    # -- inputs -----------------------------------------------------------
    def set_system_generic(self, hs_pot, rchol, H1, ecore, na, nb):
        H1 = numpy.asarray(H1, dtype=numpy.complex128)
        M = H1.shape[-1]
        hs_pot = numpy.asarray(hs_pot, dtype=numpy.float64)
        self.kind, self.M, self.K, self.na, self.nb = 'generic', M, hs_pot.shape[1], na, nb
        self._sys = dict(hs_pot=hs_pot, rchol=numpy.asarray(rchol, dtype=numpy.complex128), H1=H1,
                         ecore=float(numpy.real(ecore)))

    def set_system_hubbard(self, T, U, na, nb):
        T = numpy.asarray(T, dtype=numpy.complex128)
        M = T.shape[-1]
        self.kind, self.M, self.K, self.na, self.nb = 'hubbard', M, M, na, nb
        self._sys = dict(U=float(U), H1=T)

    def set_system_ueg(self, iA, iB, ikpq_i, ikpq_kpq, ipmq_i, ipmq_pmq, vqvec, vol, H1diag, ecore, na, nb):
        H1diag = numpy.asarray(H1diag, dtype=numpy.float64)
        self.kind, self.M, self.K, self.na, self.nb = 'ueg', H1diag.shape[-1], 2 * iA.shape[1], na, nb
        self._sys = dict(iA=iA.tocsc(), iB=iB.tocsc(), ikpq_i=ikpq_i, ikpq_kpq=ikpq_kpq, ipmq_i=ipmq_i,
                         ipmq_pmq=ipmq_pmq, vqvec=numpy.asarray(vqvec), vol=float(vol), H1diag=H1diag,
                         ecore=float(ecore))

    def set_trial(self, psi):
        self.ndet = 1
        self._psi = numpy.array(psi, dtype=numpy.complex128)
        assert self._psi.shape == (self.M, self.na + self.nb)
        self._build_model()

    def set_trial_multi(self, psi, coeffs, rchol):
        raise NotImplementedError("the stand-in covers single-determinant trials")

    def set_propagator(self, BH1, mf_shift, dt, exp_order=6, hybrid=True, force_bias=True,
                       free_projection=False, hubbard_spin=False):
        self._prop = dict(BH1=numpy.array(BH1, dtype=numpy.complex128), mf_shift=numpy.array(mf_shift),
                          dt=float(dt), exp_order=int(exp_order))
        self._hybrid, self._free = bool(hybrid), bool(free_projection)
        if not force_bias and not free_projection:
            raise NotImplementedError("force_bias: False without free projection")
        self._spin = bool(hubbard_spin)
        self.nv = 2 if (self.kind == 'hubbard' and hubbard_spin) else 1
        self._build_model()

    def _build_model(self):
        prop = getattr(self, '_prop', None)
        psi = getattr(self, '_psi', None)
        if psi is None:
            return
        if prop is None:        # overlaps / Green's functions only need the trial
            prop = dict(BH1=numpy.array([numpy.eye(self.M)] * 2, dtype=numpy.complex128),
                        mf_shift=numpy.zeros(self.K), dt=1.0, exp_order=6)
        kind = self.kind
        if kind == 'hubbard' and getattr(self, '_spin', False):
            kind = 'hubbard_spin'
        self._model = ref.RefModel(kind, self.M, self.na, self.nb, psi, prop['BH1'], prop['mf_shift'], prop['dt'],
                                   exp_order=prop['exp_order'], **self._sys)
        self._apply_log_shift()

    def _apply_log_shift(self):
        on, ls, ds = self._log_shift
        if self._model is not None:
            self._model.log_shift = ls if on else 0.0
            self._model.detR_shift = ds if on else 0.0

    # -- walkers ----------------------------------------------------------
    def walkers_alloc(self, nw):
        self.nw = int(nw)
        z = numpy.zeros((self.M, self.na + self.nb), dtype=numpy.complex128)
        self._w = [dict(phi=z.copy(), weight=1.0, unscaled_weight=1.0, ot=1.0 + 0j, ovlp=1.0 + 0j, hybrid_energy=0.0,
                        total_weight=0.0, detR=1.0, phase=1.0 + 0j, eloc=0.0, log_detR=0.0) for _ in range(self.nw)]
        self._ghalf = numpy.zeros((self.nw, self.na + self.nb, self.M), dtype=numpy.complex128)
        self._G = numpy.zeros((self.nw, 2, self.M, self.M), dtype=numpy.complex128)
        self._energy = numpy.zeros((self.nw, 3), dtype=numpy.complex128)
        self._xbar = numpy.zeros((self.nw, self.K), dtype=numpy.complex128)
        self._total_weight = float(self.nw)

    _KEYS = {L.F_WEIGHT: 'weight', L.F_UNSCALED_WEIGHT: 'unscaled_weight', L.F_OT: 'ot',
             L.F_HYBRID_ENERGY: 'hybrid_energy', L.F_PHASE: 'phase', L.F_DETR: 'detR', L.F_ELOC: 'eloc',
             L.F_LOG_DETR: 'log_detR'}
    _REAL = (L.F_WEIGHT, L.F_UNSCALED_WEIGHT, L.F_DETR, L.F_LOG_DETR)

    def set(self, field, values, first=0):
        values = numpy.asarray(values)
        if field == L.F_PHI:
            values = values.reshape((-1, self.M, self.na + self.nb))
            for j, v in enumerate(values):
                self._w[first + j]['phi'] = numpy.array(v, dtype=numpy.complex128)
        elif field == L.F_GHALF:
            values = values.reshape((-1,) + self._ghalf.shape[1:])
            self._ghalf[first:first + len(values)] = values
        elif field == L.F_G:
            values = values.reshape((-1,) + self._G.shape[1:])
            self._G[first:first + len(values)] = values
        else:
            key = self._KEYS[field]
            for j, v in enumerate(numpy.atleast_1d(values)):
                self._w[first + j][key] = float(numpy.real(v)) if field in self._REAL else complex(v)
                if key == 'ot':
                    self._w[first + j]['ovlp'] = complex(v)

    def get(self, field, first=0, count=None):
        count = self.nw - first if count is None else count
        rows = range(first, first + count)
        if field == L.F_PHI:
            return numpy.array([self._w[i]['phi'] for i in rows], dtype=numpy.complex128)
        if field == L.F_GHALF:
            return self._ghalf[first:first + count].copy()
        if field == L.F_G:
            return self._G[first:first + count].copy()
        if field == L.F_XBAR:
            return self._xbar[first:first + count].copy()
        if field == L.F_ENERGY:
            return self._energy[first:first + count].copy()
        key = self._KEYS[field]
        dt = numpy.float64 if field in self._REAL else numpy.complex128
        return numpy.array([self._w[i][key] for i in rows], dtype=dt)

    # -- hot path ---------------------------------------------------------
    def greens(self, want_G=False, fetch=True):
        self._log('greens')
        m = self._model
        det = numpy.zeros(self.nw, dtype=numpy.complex128)
        ls = numpy.exp(getattr(m, 'log_shift', 0.0))
        for i, w in enumerate(self._w):
            ov, gh, G = m.greens(w['phi'])
            det[i] = ov * ls                # the device hands back the plain determinant (WalkerView applies the shift)
            self._ghalf[i, :self.na] = gh[0]
            self._ghalf[i, self.na:] = gh[1]
            self._G[i] = G
        return det if fetch else None

    def calc_overlap(self, fetch=True):
        self._log('calc_overlap')
        m = self._model
        ls = numpy.exp(getattr(m, 'log_shift', 0.0))
        out = numpy.array([m.overlap(w['phi']) * ls for w in self._w], dtype=numpy.complex128)
        return out if fetch else None

    def propagate(self, xi, eshift):
        self._log('propagate')
        if xi is None:
            raise NotImplementedError("the stand-in has no device random stream: pass the fields")
        m = self._model
        for i, w in enumerate(self._w):
            if abs(w['weight']) > 1e-8:                                    # qmc/afqmc.py:232
                if self._free:
                    ref.propagate_walker_free(m, w, numpy.asarray(xi[i]), eshift)
                else:
                    nfb, nhe = ref.propagate_walker_phaseless(m, w, numpy.asarray(xi[i]), eshift, self._hybrid)
                    self._ntrig[0] += nfb
                    self._ntrig[1] += nhe
            if self._cap > 0:
                tw = self._cap_total if self._cap_total >= 0 else self._total_weight
                if abs(w['weight']) > tw * self._cap:
                    w['weight'] = tw * self._cap

    def reortho(self, fetch=True):
        self._log('reortho')
        m = self._model
        out = numpy.zeros(self.nw)
        for i, w in enumerate(self._w):
            detR = ref.reortho(w['phi'], self.na, self.nb, getattr(m, 'detR_shift', 0.0))
            w['detR'] = detR
            w['log_detR'] += numpy.log(detR)
            w['ot'] = w['ot'] / detR
            w['ovlp'] = w['ot']
            if self._free:                                                 # walkers/handler.py:178-181
                w['weight'] *= abs(detR)
            out[i] = detR
        return out if fetch else None

    def set_log_shift(self, on, log_shift=0.0, detR_shift=0.0):
        self._log_shift = (bool(on), float(log_shift), float(detR_shift))
        self._apply_log_shift()

    def log_ovlp_sums(self):
        return numpy.array([sum(abs(w['ot']) for w in self._w), sum(abs(w['detR']) for w in self._w),
                            sum(abs(w['log_detR']) for w in self._w)], dtype=numpy.float64)

    def local_energy(self, fetch=True):
        """Energies of the Green's functions currently held (afq_local_energy reads Ghalf / G as stored)."""
        self._log('local_energy')
        m = self._model
        na = self.na
        for i in range(self.nw):
            gh = [self._ghalf[i, :na], self._ghalf[i, na:]]
            G = self._G[i]
            if self.kind != 'ueg':          # the library's Generic / Hubbard energies read the half-rotated form only
                G = numpy.array([m.psi[:, :na].conj().dot(gh[0]), m.psi[:, na:].conj().dot(gh[1])])
            self._energy[i] = numpy.array(m.local_energy(G, gh), dtype=numpy.complex128)
        return self._energy.copy() if fetch else None

    def force_bias(self):
        self._log('force_bias')
        m = self._model
        for i in range(self.nw):
            gh = [self._ghalf[i, :self.na], self._ghalf[i, self.na:]]
            self._xbar[i] = m.force_bias(gh, self._G[i])
        return self._xbar.copy()

    def vhs(self, xs):
        self._log('vhs')
        out = numpy.zeros((self.nw, self.nv, self.M, self.M), dtype=numpy.complex128)
        for i in range(self.nw):
            v = numpy.asarray(self._model.vhs(numpy.asarray(xs[i])))
            out[i] = v if v.ndim == 3 else v[None]
        return out

    # -- driver glue ------------------------------------------------------
    def cap_weights(self, frac, total_weight):
        tw = total_weight if total_weight >= 0 else self._total_weight
        for w in self._w:
            if abs(w['weight']) > tw * frac:
                w['weight'] = tw * frac

    def set_weight_cap(self, frac, total_weight=-1.0):
        self._cap, self._cap_total = float(frac), float(total_weight)

    def popcontrol_comb(self, r, target, fetch=True):
        self._log('popcontrol_comb')
        weights = numpy.array([abs(w['weight']) for w in self._w])
        total = float(sum(weights))
        if total < 1e-8:
            raise L.AfqError(L.AFQ_EWEIGHT, "total weight below 1e-8")
        pix = ref.pop_control(self._model, self._w, int(target), r)        # the shifts are updated by Walkers itself
        self._total_weight = total
        return (numpy.asarray(pix, dtype=numpy.int32), total) if fetch else (None, None)

    def scale_weights(self, scale):
        for w in self._w:
            w['unscaled_weight'] = w['weight']
            w['weight'] = w['weight'] / scale

    def reset_weights(self):
        for w in self._w:
            w['weight'] = 1.0

    def copy_walker(self, src, dst):
        for k, v in self._w[src].items():
            self._w[dst][k] = numpy.array(v, copy=True) if isinstance(v, numpy.ndarray) else v

    def estimates_update(self, eval_energy):
        self._log('estimates_update')
        step, freq = (0, 1) if eval_energy else (1, 2)
        if eval_energy:
            self.greens(fetch=False)        # an energy step leaves the walkers' Green's functions behind, as the library does
        ref.mixed_update(self._model, self._est, self._w, step, freq, self._free)

    def estimates_get(self, zero=False):
        out = self._est.copy()
        if zero:
            self._est[:] = 0
        return out

    def estimates_rdm(self, on=True):
        if on:
            raise NotImplementedError("one_rdm accumulation is not part of the stand-in")

    # -- discrete Hirsch fields (afq_set_propagator_hirsch / afq_hirsch_kinetic / _two_body / _finish), constrained path:
    #    propagation/hubbard.py:285-312 split where the host-uniform path of the library splits it
    def set_propagator_hirsch(self, bt2, dt, charge_decomposition=False):
        self.nv = 1
        self._model = ref.HirschModel(self._sys['H1'], self._sys['U'], self._psi, self.na, self.nb, float(dt),
                                      bool(charge_decomposition))
        assert numpy.allclose(self._model.bt2, bt2, rtol=1e-13, atol=1e-15)
        self._model.single_site = True
        self._hybrid, self._free = False, False
        self._apply_log_shift()

    def hirsch_single_site(self, on=True):
        self._model.single_site = bool(on)

    def hirsch_free_projection(self, on=True):
        if on:
            raise NotImplementedError("free projection with discrete fields is not part of the stand-in")

    def hirsch_kinetic(self):
        self._log('hirsch_kinetic')
        self._live = [abs(w['weight']) > 1e-8 for w in self._w]                 # qmc/afqmc.py:232, at the start of the sweep
        for w, live in zip(self._w, self._live):
            if live and abs(w['weight']) > 0:
                ref.hirsch_kinetic_importance_sampling(self._model, w)

    def hirsch_two_body(self, u):
        self._log('propagate')                                                  # the one batched site-loop launch per step
        u = numpy.asarray(u, dtype=numpy.float64).reshape(self.nw, self.M)
        fields = numpy.zeros((self.nw, self.M), dtype=numpy.int32)
        used = numpy.zeros(self.nw, dtype=numpy.int32)
        for i, (w, live) in enumerate(zip(self._w, self._live)):
            if not (live and abs(w['weight']) > 0):
                continue
            it = iter(u[i])
            count = [0]

            def uniform():
                count[0] += 1
                return next(it)
            if self._model.single_site:
                f = ref.hirsch_two_body_single_site(self._model, w, uniform)
            else:
                f = ref.hirsch_two_body_direct(self._model, w, uniform)
            fields[i, :len(f)] = f
            used[i] = count[0]
        return fields, used

    def hirsch_finish(self, eshift):
        for w, live in zip(self._w, self._live):
            if not live:
                continue
            if abs(numpy.real(w['weight'])) > 0:
                ref.hirsch_kinetic_importance_sampling(self._model, w)
            w['weight'] *= numpy.exp(self._model.dt * eshift)
            w['ovlp'] = w['ot']

    # -- back-propagation (afq_bp_configure / afq_bp_steps / afq_bp_update): the oracle's field history per walker
    def bp_configure(self, nbp):
        self.nbp = int(nbp)
        for w in self._w:
            w['bp'] = ref.bp_new(self.K, self.nbp)
            w['phi_old'] = w['phi'].copy()

    def bp_steps(self):
        return numpy.array([w['bp']['step'] for w in self._w], dtype=numpy.int32)

    def bp_update(self, phi_bp0, nstblz, restore_weights=None, eval_energy=False, reset=True):
        self._log('bp_update')
        est = numpy.zeros(4 + 2 * self.M * self.M, dtype=numpy.complex128)
        ref.bp_update(self._model, self._w, nstblz, est, restore_weights, init=numpy.asarray(phi_bp0),
                      eval_energy=eval_energy, reset=reset)
        return est[:3], est[3], est[4:].reshape(2, self.M, self.M)

    def counters(self, reset=False):
        out = self._ntrig.copy()
        if reset:
            self._ntrig[:] = 0
        return out

    def rng_seed(self, seed, stream=0):
        raise NotImplementedError("the stand-in has no device random stream")


def install(monkeypatch=None):
    """pauxy_amd.context builds its handle with the name ``AfqDevice`` of its own namespace: rebinding that name is the
    whole installation.  With a pytest ``monkeypatch`` the binding is undone at the end of the test."""
    import pauxy_amd.context as ctx
    if monkeypatch is not None:
        monkeypatch.setattr(ctx, 'AfqDevice', OracleDevice)
        monkeypatch.setattr(ctx, '_contexts', {})
    else:
        ctx.AfqDevice = OracleDevice
        ctx._contexts.clear()
    return OracleDevice
