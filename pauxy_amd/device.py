"""Thin numpy-facing wrapper over the C ABI: one ``AfqDevice`` per GPU.

Every array crossing this boundary is a C-contiguous numpy array; complex128
arrays are passed as their interleaved (re, im) storage, exactly what
``include/afqmc_hip.h`` documents.  All arithmetic happens in the library.
"""
import ctypes

import numpy

from pauxy_amd import _lib as L


def _c128(a, shape=None):
    a = numpy.ascontiguousarray(a, dtype=numpy.complex128)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def _f64(a, shape=None):
    a = numpy.ascontiguousarray(a, dtype=numpy.float64)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def _i64(a):
    return numpy.ascontiguousarray(a, dtype=numpy.int64)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _flatten_ragged(lists):
    off = numpy.zeros(len(lists) + 1, dtype=numpy.int64)
    off[1:] = numpy.cumsum([len(x) for x in lists])
    flat = numpy.concatenate([numpy.asarray(x, dtype=numpy.int64) for x in lists]) if off[-1] else \
        numpy.zeros(0, dtype=numpy.int64)
    return off, _i64(flat)


class AfqDevice(object):
    def __init__(self, device_id=0):
        self.lib = L.load()
        self.h = ctypes.c_void_p()
        rc = self.lib.afq_create(int(device_id), ctypes.byref(self.h))
        if rc != 0:
            raise L.AfqError(rc, "afq_create failed (no usable MI355X device %d?)" % device_id)
        self.device_id = device_id
        self.kind = None
        self.nw = 0

    # -- plumbing ---------------------------------------------------------
    def _ck(self, rc):
        if rc != 0:
            raise L.AfqError(rc, self.lib.afq_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.afq_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._ck(self.lib.afq_sync(self.h))

    # -- inputs -----------------------------------------------------------
    def set_system_generic(self, hs_pot, rchol, H1, ecore, na, nb):
        H1 = _c128(H1)
        M = H1.shape[-1]
        hs_pot = _f64(hs_pot)
        K = hs_pot.shape[1]
        assert hs_pot.shape == (M * M, K)
        rchol = _c128(rchol, ((na + nb) * M, K))
        self._ck(self.lib.afq_set_system_generic(self.h, M, K, na, nb, _p(hs_pot), _p(rchol), _p(H1),
                                                 float(numpy.real(ecore))))
        self.kind, self.M, self.K, self.na, self.nb = 'generic', M, K, na, nb

    def set_system_hubbard(self, T, U, na, nb):
        T = _c128(T)
        M = T.shape[-1]
        self._ck(self.lib.afq_set_system_hubbard(self.h, M, na, nb, float(U), _p(T)))
        self.kind, self.M, self.K, self.na, self.nb = 'hubbard', M, M, na, nb

    def set_system_ueg(self, iA, iB, ikpq_i, ikpq_kpq, ipmq_i, ipmq_pmq, vqvec, vol, H1diag, ecore, na, nb):
        iA = iA.tocsc()
        iB = iB.tocsc()
        iA.sort_indices()
        iB.sort_indices()
        H1diag = _f64(H1diag)
        M = H1diag.shape[-1]
        nq = iA.shape[1]
        koff, ki = _flatten_ragged(ikpq_i)
        _, kk = _flatten_ragged(ikpq_kpq)
        poff, pi = _flatten_ragged(ipmq_i)
        _, pp = _flatten_ragged(ipmq_pmq)
        arrs = [_i64(iA.indptr), _i64(iA.indices), _c128(iA.data), _i64(iB.indptr), _i64(iB.indices),
                _c128(iB.data), koff, ki, kk, poff, pi, pp, _f64(vqvec)]
        self._ck(self.lib.afq_set_system_ueg(self.h, M, nq, na, nb, *[_p(a) for a in arrs], float(vol),
                                             _p(H1diag), float(ecore)))
        self.kind, self.M, self.K, self.na, self.nb = 'ueg', M, 2 * nq, na, nb

    def set_trial(self, psi):
        self.ndet = 1
        psi = _c128(psi, (self.M, self.na + self.nb))
        self._ck(self.lib.afq_set_trial(self.h, _p(psi)))

    def inverse_overlap(self):
        """(O^-1 [nw, 2, nmax, nmax], overlap [nw]) with O = phi_s^T conj(psi_s)."""
        nmax = max(self.na, self.nb)
        out = numpy.zeros((self.nw, 2, nmax, nmax), dtype=numpy.complex128)
        ov = numpy.zeros(self.nw, dtype=numpy.complex128)
        self._ck(self.lib.afq_inverse_overlap(self.h, _p(out), _p(ov)))
        return out, ov

    # ---- discrete Hirsch propagator
    def set_propagator_hirsch(self, bt2, dt, charge_decomposition=False):
        bt2 = _c128(bt2, (2, self.M, self.M))
        self._ck(self.lib.afq_set_propagator_hirsch(self.h, _p(bt2), float(dt), int(bool(charge_decomposition))))
        self.nv = 1

    def propagate_hirsch(self, eshift):
        self._ck(self.lib.afq_propagate_hirsch(self.h, float(numpy.real(eshift))))

    def hirsch_single_site(self, on=True):
        """on=False: two_body_direct (propagation/hubbard.py:222-275) instead of the M single-site updates."""
        self._ck(self.lib.afq_hirsch_single_site(self.h, 1 if on else 0))

    def hirsch_free_projection(self, on=True):
        self._ck(self.lib.afq_hirsch_free_projection(self.h, 1 if on else 0))

    def propagate_hirsch_free(self, u, eshift, fetch_fields=False):
        """propagation/hubbard.py:303-343 for every walker with |weight| > 1e-8; u [nw, M] or None (device stream)."""
        u_ = None if u is None else _f64(u, (self.nw, self.M))
        fields = numpy.zeros((self.nw, self.M), dtype=numpy.int32) if fetch_fields else None
        self._ck(self.lib.afq_propagate_hirsch_free(self.h, _p(u_), _p(fields), float(eshift)))
        return fields

    def hirsch_kinetic(self):
        self._ck(self.lib.afq_hirsch_kinetic(self.h))

    def hirsch_two_body(self, u):
        u = numpy.ascontiguousarray(u, dtype=numpy.float64).reshape(self.nw, self.M)
        fields = numpy.zeros((self.nw, self.M), dtype=numpy.int32)
        used = numpy.zeros(self.nw, dtype=numpy.int32)
        self._ck(self.lib.afq_hirsch_two_body(self.h, _p(u), _p(fields), _p(used)))
        return fields, used

    def hirsch_finish(self, eshift):
        self._ck(self.lib.afq_hirsch_finish(self.h, float(numpy.real(eshift))))

    # ---- back-propagation
    def bp_configure(self, nbp):
        self._ck(self.lib.afq_bp_configure(self.h, int(nbp)))
        self.nbp = int(nbp)

    def bp_steps(self):
        out = numpy.zeros(self.nw, dtype=numpy.int32)
        self._ck(self.lib.afq_bp_steps(self.h, _p(out)))
        return out

    def local_energy_full_g(self, G):
        """(E, E1b, E2b) of Green's functions G[n, 2, M, M] by the full-G Cholesky energy."""
        G = _c128(G)
        n = G.shape[0]
        G = _c128(G, (n, 2, self.M, self.M))
        out = numpy.zeros((n, 3), dtype=numpy.complex128)
        self._ck(self.lib.afq_local_energy_full_g(self.h, _p(G), n, _p(out)))
        return out

    def bp_update(self, phi_bp0, nstblz, restore_weights=None, eval_energy=False, reset=True):
        """-> (energies_sum[3], denominator, G_bp_sum[2, M, M]); restore_weights in (None, 'partial', 'full')."""
        mode = {None: 0, 'partial': 1, 'full': 2}.get(restore_weights, 1)
        phi0 = _c128(phi_bp0, (self.M, self.na + self.nb))
        out = numpy.zeros(4 + 2 * self.M * self.M, dtype=numpy.complex128)
        self._ck(self.lib.afq_bp_update(self.h, _p(phi0), int(nstblz), mode, int(bool(eval_energy)), int(bool(reset)),
                                        _p(out)))
        return out[:3], out[3], out[4:].reshape(2, self.M, self.M)

    def set_trial_multi(self, psi, coeffs, rchol):
        """psi [ndet, M, na+nb], coeffs [ndet], rchol [ndet * (na+nb) M, K] (stacked per determinant)."""
        nt = self.na + self.nb
        psi = _c128(psi)
        ndet = psi.shape[0]
        psi = _c128(psi, (ndet, self.M, nt))
        coeffs = _c128(coeffs, (ndet,))
        rchol = _c128(rchol, (ndet * nt * self.M, self.K))
        self._ck(self.lib.afq_set_trial_multi(self.h, ndet, _p(psi), _p(coeffs), _p(rchol)))
        self.ndet = ndet

    def det_weights(self):
        """conj(c_d) <D_d|phi_w> of the last Green's function / overlap evaluation, [nw, ndet]."""
        out = numpy.zeros((self.nw, self.ndet), dtype=numpy.complex128)
        self._ck(self.lib.afq_walkers_det_weights(self.h, _p(out)))
        return out

    def set_propagator(self, BH1, mf_shift, dt, exp_order=6, hybrid=True, force_bias=True,
                       free_projection=False, hubbard_spin=False):
        BH1 = _c128(BH1, (2, self.M, self.M))
        mf = _c128(mf_shift, (self.K,))
        flags = ((L.AFQ_PROP_HYBRID if hybrid else 0) | (L.AFQ_PROP_FORCE_BIAS if force_bias else 0) |
                 (L.AFQ_PROP_FREE_PROJECTION if free_projection else 0) |
                 (L.AFQ_PROP_HUBBARD_SPIN if hubbard_spin else 0))
        self._ck(self.lib.afq_set_propagator(self.h, _p(BH1), _p(mf), float(dt), int(exp_order), flags))
        nv = ctypes.c_int()
        self._ck(self.lib.afq_vhs_count(self.h, ctypes.byref(nv)))
        self.nv = nv.value

    # -- walkers ----------------------------------------------------------
    def walkers_alloc(self, nw):
        self._ck(self.lib.afq_walkers_alloc(self.h, int(nw)))
        self.nw = int(nw)

    def _field_spec(self, field):
        M, nt, K = self.M, self.na + self.nb, self.K
        return {
            L.F_PHI: ((M, nt), numpy.complex128), L.F_WEIGHT: ((), numpy.float64),
            L.F_UNSCALED_WEIGHT: ((), numpy.float64), L.F_OT: ((), numpy.complex128),
            L.F_HYBRID_ENERGY: ((), numpy.complex128), L.F_PHASE: ((), numpy.complex128),
            L.F_DETR: ((), numpy.float64), L.F_ELOC: ((), numpy.complex128),
            L.F_GHALF: ((nt, M), numpy.complex128), L.F_G: ((2, M, M), numpy.complex128),
            L.F_XBAR: ((K,), numpy.complex128), L.F_XSHIFTED: ((K,), numpy.complex128),
            L.F_ENERGY: ((3,), numpy.complex128), L.F_LOG_DETR: ((), numpy.float64),
        }[field]

    def set(self, field, values, first=0):
        shape, dt = self._field_spec(field)
        values = numpy.ascontiguousarray(values, dtype=dt)
        if values.shape == shape:
            values = values.reshape((1,) + shape)
        assert values.shape[1:] == shape, (values.shape, shape)
        self._ck(self.lib.afq_walkers_set(self.h, field, _p(values), int(first), values.shape[0]))

    def get(self, field, first=0, count=None):
        shape, dt = self._field_spec(field)
        count = self.nw - first if count is None else count
        out = numpy.empty((count,) + shape, dtype=dt)
        self._ck(self.lib.afq_walkers_get(self.h, field, _p(out), int(first), int(count)))
        return out

    # -- hot path ---------------------------------------------------------
    def greens(self, want_G=False, fetch=True):
        out = numpy.empty(self.nw, dtype=numpy.complex128) if fetch else None
        self._ck(self.lib.afq_greens(self.h, int(bool(want_G)), _p(out)))
        return out

    def calc_overlap(self, fetch=True):
        out = numpy.empty(self.nw, dtype=numpy.complex128) if fetch else None
        self._ck(self.lib.afq_calc_overlap(self.h, _p(out)))
        return out

    def propagate(self, xi, eshift):
        if xi is not None:
            xi = _f64(xi, (self.nw, self.K))
        es = complex(eshift)
        self._ck(self.lib.afq_propagate(self.h, _p(xi), es.real, es.imag))

    def propagate_begin(self, xi):
        """Everything of a step that does not read the energy shift (afq_propagate_begin)."""
        if xi is not None:
            xi = _f64(xi, (self.nw, self.K))
        self._ck(self.lib.afq_propagate_begin(self.h, _p(xi)))

    def propagate_finish(self, eshift):
        es = complex(eshift)
        self._ck(self.lib.afq_propagate_finish(self.h, es.real, es.imag))

    def reortho(self, fetch=True):
        out = numpy.empty(self.nw, dtype=numpy.float64) if fetch else None
        self._ck(self.lib.afq_reortho(self.h, _p(out)))
        return out

    def set_log_shift(self, on, log_shift=0.0, detR_shift=0.0):
        """use_log_shift (walkers/handler.py:45): walker.log_shift / walker.detR_shift of every walker."""
        self._ck(self.lib.afq_set_log_shift(self.h, int(bool(on)), float(log_shift), float(detR_shift)))

    def log_ovlp_sums(self):
        """sums of |ot|, |detR|, |log_detR| over this rank's walkers (walkers/handler.py:457-462)."""
        out = numpy.empty(3, dtype=numpy.float64)
        self._ck(self.lib.afq_log_ovlp_sums(self.h, _p(out)))
        return out

    def local_energy(self, fetch=True):
        out = numpy.empty((self.nw, 3), dtype=numpy.complex128) if fetch else None
        self._ck(self.lib.afq_local_energy(self.h, _p(out)))
        return out

    def force_bias(self):
        out = numpy.empty((self.nw, self.K), dtype=numpy.complex128)
        self._ck(self.lib.afq_force_bias(self.h, _p(out)))
        return out

    def shift_fields(self, xi, xbar):
        xi = _f64(xi, (self.nw, self.K))
        xbar = _c128(xbar, (self.nw, self.K))
        xs = numpy.empty((self.nw, self.K), dtype=numpy.complex128)
        cmf = numpy.empty(self.nw, dtype=numpy.complex128)
        cfb = numpy.empty(self.nw, dtype=numpy.complex128)
        self._ck(self.lib.afq_shift_fields(self.h, _p(xi), _p(xbar), _p(xs), _p(cmf), _p(cfb)))
        return xs, cmf, cfb

    def vhs(self, xs):
        xs = _c128(xs, (self.nw, self.K))
        out = numpy.empty((self.nw, self.nv, self.M, self.M), dtype=numpy.complex128)
        self._ck(self.lib.afq_vhs(self.h, _p(xs), _p(out)))
        return out

    def apply_exponential(self, vhs):
        vhs = _c128(vhs, (self.nw, self.nv, self.M, self.M))
        self._ck(self.lib.afq_apply_exponential(self.h, _p(vhs)))

    def kinetic(self):
        self._ck(self.lib.afq_kinetic(self.h))

    # -- driver glue ------------------------------------------------------
    def cap_weights(self, frac, total_weight):
        self._ck(self.lib.afq_cap_weights(self.h, float(frac), float(total_weight)))

    def set_weight_cap(self, frac, total_weight=-1.0):
        """Cap applied by propagate() itself from now on (frac <= 0: off); see afq_set_weight_cap."""
        self._ck(self.lib.afq_set_weight_cap(self.h, float(frac), float(total_weight)))

    def popcontrol_comb(self, r, target, fetch=True):
        """Single rank, or a collective over the ranks of comm_init (parent_ix is then the global comb)."""
        if not fetch:        # asynchronous: the total weight stays on the device (cap_weights(frac, -1))
            self._ck(self.lib.afq_popcontrol_comb(self.h, float(r), float(target), None, None))
            return None, None
        pix = numpy.zeros(self.nw * getattr(self, 'comm_size', 1), dtype=numpy.int32)
        tw = ctypes.c_double(0.0)
        self._ck(self.lib.afq_popcontrol_comb(self.h, float(r), float(target), _p(pix), ctypes.byref(tw)))
        return pix, tw.value

    def scale_weights(self, scale):
        self._ck(self.lib.afq_walkers_scale_weights(self.h, float(scale)))

    def reset_weights(self):
        self._ck(self.lib.afq_walkers_reset_weights(self.h))

    def copy_walker(self, src, dst):
        self._ck(self.lib.afq_walkers_copy(self.h, int(src), int(dst)))

    def pack_bytes(self):
        n = ctypes.c_int64()
        self._ck(self.lib.afq_walker_pack_bytes(self.h, ctypes.byref(n)))
        return n.value

    def pack(self, iw, dev_ptr):
        self._ck(self.lib.afq_walker_pack(self.h, int(iw), ctypes.c_void_p(dev_ptr)))

    def unpack(self, iw, dev_ptr):
        self._ck(self.lib.afq_walker_unpack(self.h, int(iw), ctypes.c_void_p(dev_ptr)))

    def estimates_update(self, eval_energy):
        self._ck(self.lib.afq_estimates_update(self.h, int(bool(eval_energy))))

    def estimates_rdm(self, on=True):
        self._ck(self.lib.afq_estimates_rdm(self.h, int(bool(on))))

    def estimates_rdm_get(self, zero=False):
        out = numpy.empty((2, self.M, self.M), dtype=numpy.float64)
        self._ck(self.lib.afq_estimates_rdm_get(self.h, _p(out), int(bool(zero))))
        return out

    def estimates_get(self, zero=False):
        out = numpy.empty(10, dtype=numpy.complex128)
        self._ck(self.lib.afq_estimates_get(self.h, _p(out), int(bool(zero))))
        return out

    def estimates_fuse_next(self):
        """The next propagate / propagate_finish takes the step's estimator terms along (no estimates_update(False))."""
        self._ck(self.lib.afq_estimates_fuse_next(self.h))

    def estimates_update_publish(self, eval_energy, zero=False):
        """estimates_update + estimates_get_begin in one summation launch (the step that ends a block)."""
        self._ck(self.lib.afq_estimates_update_publish(self.h, int(bool(eval_energy)), int(bool(zero))))

    def estimates_get_begin(self, zero=False):
        """Enqueue the fetch; work enqueued afterwards keeps running while estimates_get_end waits for the sums."""
        self._ck(self.lib.afq_estimates_get_begin(self.h, int(bool(zero))))

    def estimates_get_end(self):
        out = numpy.empty(10, dtype=numpy.complex128)
        self._ck(self.lib.afq_estimates_get_end(self.h, _p(out)))
        return out

    # -- library-owned communicator -----------------------------------------
    def comm_available(self):
        """librccl loadable in this process (no collective, no GPU work)."""
        return bool(self.lib.afq_comm_available())

    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        rc = self.lib.afq_comm_unique_id(buf)
        if rc != 0:
            raise L.AfqError(rc, "afq_comm_unique_id failed (librccl not loadable?)")
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        self._ck(self.lib.afq_comm_init(self.h, ctypes.c_char_p(unique_id), int(rank), int(nranks)))
        self.comm_rank, self.comm_size = int(rank), int(nranks)

    def comm_init_ipc(self, rank, nranks, allgather):
        """Communicator over mapped peer windows, no RCCL.  ``allgather(send_bytes) -> bytes of every rank
        concatenated`` is the caller's all-gather (its MPI / torch.distributed communicator): the bootstrap of the
        window handles, called on the host whenever the library (re)makes its windows."""
        def thunk(send, recv, nbytes, _user):
            try:
                out = allgather(ctypes.string_at(send, nbytes))
                if len(out) != nbytes * nranks:
                    return 1
                ctypes.memmove(recv, out, len(out))
                return 0
            except Exception:                                   # noqa: BLE001  (must not unwind through C)
                return 1
        self._ipc_thunk = L.ALLGATHER_FN(thunk)                 # kept alive as long as the communicator
        self._ck(self.lib.afq_comm_init_ipc(self.h, int(rank), int(nranks), self._ipc_thunk, None))
        self.comm_rank, self.comm_size = int(rank), int(nranks)

    def comm_destroy(self):
        self._ck(self.lib.afq_comm_destroy(self.h))
        self.comm_rank, self.comm_size = 0, 1

    def comm_set_transport(self, window):
        self._ck(self.lib.afq_comm_set_transport(self.h, 1 if window else 0))

    def comm_set_capacity(self, cap):
        self._ck(self.lib.afq_comm_set_capacity(self.h, int(cap)))

    def comm_set_timeout(self, seconds):
        """Wait budget of the kernels that wait for a peer (default 300 s; AFQ_COMM_TIMEOUT_S)."""
        self._ck(self.lib.afq_comm_set_timeout(self.h, float(seconds)))

    def comm_probe(self):
        """Collective known-answer round (all-gather, one slot per peer, all-reduce); raises AfqError(AFQ_ECOMM)."""
        bad = numpy.zeros(3, dtype=numpy.int64)
        self._ck(self.lib.afq_comm_probe(self.h, _p(bad)))
        return bad

    def comm_stats(self):
        out = numpy.zeros(L.AFQ_COMM_NSTATS, dtype=numpy.int64)
        self._ck(self.lib.afq_comm_stats(self.h, _p(out)))
        st = dict(zip(['max_transfer', 'events', 'capacity', 'overflow', 'rank', 'size', 'walkers_sent', 'bytes_sent',
                       'window', 'error', 'kind', 'window_memory'], out.tolist()))
        st['transport'] = 'peer windows' if st['window'] else 'ncclSend/ncclRecv'
        st['kind'] = {0: 'in-process', 1: 'rccl', 2: 'ipc'}.get(st['kind'], st['kind'])
        return st

    def estimates_allreduce(self, buf=None):
        """Sum over the ranks of the communicator: the device accumulators in place (buf None) or a host array."""
        if buf is None:
            self._ck(self.lib.afq_estimates_allreduce(self.h, None, 0))
            return None
        a = numpy.ascontiguousarray(buf, dtype=numpy.complex128)
        self._ck(self.lib.afq_estimates_allreduce(self.h, _p(a), a.size))
        return a

    def rng_normal(self, n):
        out = numpy.empty(int(n), dtype=numpy.float64)
        self._ck(self.lib.afq_rng_normal(self.h, _p(out), int(n)))
        return out

    def philox4x32(self, ctr_key):
        ck = numpy.ascontiguousarray(ctr_key, dtype=numpy.uint32).reshape(-1, 6)
        out = numpy.zeros((ck.shape[0], 4), dtype=numpy.uint32)
        self._ck(self.lib.afq_rng_philox4x32(self.h, _p(ck), _p(out), ck.shape[0]))
        return out

    def debug(self, sync_every_launch=False, markers=False):
        self._ck(self.lib.afq_debug(self.h, int(bool(sync_every_launch)), int(bool(markers))))

    def last_launch(self):
        """(text, queued, retired): safe to call from a watchdog thread while another thread is blocked in a sync."""
        buf = ctypes.create_string_buffer(1024)
        q, r = ctypes.c_uint64(), ctypes.c_uint64()
        self.lib.afq_last_launch(self.h, buf, 1024, ctypes.byref(q), ctypes.byref(r))
        return buf.value.decode(errors='replace'), q.value, r.value

    def rng_seed(self, seed, stream=0):
        self._ck(self.lib.afq_rng_seed(self.h, int(seed), int(stream)))

    def counters(self, reset=False, n=4):
        """[nfb_trig, nhe_trig, flagged overlap matrices, closed-deal propagator walker steps] and with n = 8 also
        [.., one-spin exchange-energy walker evaluations, one-spin Green's functions, NaN determinant overlaps taken as zero,
        closed-shell walker steps of the large-system GEMM chain] (afq_counters_ext)."""
        out = numpy.zeros(max(4, n), dtype=numpy.int64)
        self._ck(self.lib.afq_counters_ext(self.h, _p(out), int(max(4, n)), int(reset)))
        return out

    def enable_timers(self, on=True):
        self._ck(self.lib.afq_enable_timers(self.h, int(on)))

    def timers(self, reset=False):
        out = numpy.zeros(8, dtype=numpy.float64)
        self._ck(self.lib.afq_timers(self.h, _p(out), int(reset)))
        return dict(zip(['greens', 'one_body', 'force_bias', 'vhs', 'exponential', 'overlap_weight',
                         'reortho', 'energy'], out))

    def kernel_trace(self, on=True, kinds=None):
        """Event pairs around the hot kernels' launches: every kind, or only ``kinds`` (list of K_* constants)."""
        arg = int(bool(on))
        if on and kinds is not None:
            arg = sum(2 << int(k) for k in kinds)
        self._ck(self.lib.afq_kernel_trace(self.h, arg))

    def kernel_trace_stride(self, kind, stride):
        """Time only every ``stride``-th launch of ``kind`` (an event pair costs a few microseconds of pipeline bubble)."""
        self._ck(self.lib.afq_kernel_trace_stride(self.h, int(kind), int(stride)))

    def kernel_trace_get(self, kind, max_n=4096):
        """Durations [ms] of the traced launches of kernel ``kind`` (_lib.K_*)."""
        out = numpy.zeros(max_n, dtype=numpy.float64)
        n = ctypes.c_int()
        self._ck(self.lib.afq_kernel_trace_get(self.h, int(kind), _p(out), max_n, ctypes.byref(n)))
        return out[:min(n.value, max_n)]

    def launch_trace(self, on=True):
        """Event pairs around every launch, keyed by name (afq_launch_trace): a profile pass."""
        self._ck(self.lib.afq_launch_trace(self.h, int(bool(on))))

    def launch_trace_get(self):
        """{name: (launches, total ms)} of the launches recorded since launch_trace(True)."""
        cap = 256
        names = ctypes.create_string_buffer(cap * 96)
        ms = numpy.zeros(cap, dtype=numpy.float64)
        cnt = numpy.zeros(cap, dtype=numpy.int64)
        n = ctypes.c_int()
        self._ck(self.lib.afq_launch_trace_get(self.h, ctypes.cast(names, ctypes.c_void_p), len(names), _p(ms), _p(cnt),
                                               cap, ctypes.byref(n)))
        out, at = {}, 0
        for k in range(min(n.value, cap)):
            end = names.raw.index(b'\0', at)
            out[names.raw[at:end].decode()] = (int(cnt[k]), float(ms[k]))
            at = end + 1
        return out

    def kernel_issued_flops(self, kind):
        """Matrix-pipe flops of the last launch of ``kind`` (padding included, 3M products counted as 3)."""
        out = ctypes.c_double(0.0)
        self._ck(self.lib.afq_kernel_issued_flops(self.h, int(kind), ctypes.byref(out)))
        return float(out.value)

    def propagator_issued_flops(self):
        """(open, closed): matrix-pipe flops ONE walker issues in the fused propagator's open-shell / closed-shell deal."""
        a, b = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._ck(self.lib.afq_propagator_issued_flops(self.h, ctypes.byref(a), ctypes.byref(b)))
        return float(a.value), float(b.value)

    def set_exchange_algorithm(self, mode):
        """0 automatic, 1 T-intermediate (exx_kernel), 2 quadratic form (see afq_set_exchange_algorithm)."""
        self._ck(self.lib.afq_set_exchange_algorithm(self.h, int(mode)))

    def exchange_algorithm(self):
        m = ctypes.c_int()
        self._ck(self.lib.afq_exchange_algorithm(self.h, ctypes.byref(m)))
        return m.value

    def set_msd_force_bias(self, mode):
        """0 automatic, 1 one contraction per determinant, 2 determinant-averaged G (see afq_set_msd_force_bias)."""
        self._ck(self.lib.afq_set_msd_force_bias(self.h, int(mode)))

    def msd_force_bias(self):
        m = ctypes.c_int()
        self._ck(self.lib.afq_msd_force_bias(self.h, ctypes.byref(m)))
        return m.value


# -- in-process communicator: several AfqDevice objects driven by one host thread -----------------------------
def _handle_array(devs):
    return (ctypes.c_void_p * len(devs))(*[d.h.value for d in devs])


def comm_init_local(devs):
    """devs[i] becomes rank i of one communicator (afq_comm_init_local)."""
    lib = devs[0].lib
    rc = lib.afq_comm_init_local(_handle_array(devs), len(devs))
    if rc != 0:
        raise L.AfqError(rc, lib.afq_last_error(devs[0].h).decode())
    for i, d in enumerate(devs):
        d.comm_rank, d.comm_size = i, len(devs)


def popcontrol_comb_local(devs, r, target, fetch=True):
    """walkers/handler.py:225-338 over the ranks of comm_init_local -> (global parent_ix, total weight)."""
    lib = devs[0].lib
    n = len(devs)
    if not fetch:
        rc = lib.afq_popcontrol_comb_local(_handle_array(devs), n, float(r), float(target), None, None)
        pix = tw = None
    else:
        pix = numpy.zeros(n * devs[0].nw, dtype=numpy.int32)
        tw = ctypes.c_double(0.0)
        rc = lib.afq_popcontrol_comb_local(_handle_array(devs), n, float(r), float(target), _p(pix), ctypes.byref(tw))
        tw = tw.value
    if rc != 0:
        raise L.AfqError(rc, lib.afq_last_error(devs[0].h).decode())
    return pix, tw


def estimates_allreduce_local(devs):
    lib = devs[0].lib
    rc = lib.afq_estimates_allreduce_local(_handle_array(devs), len(devs))
    if rc != 0:
        raise L.AfqError(rc, lib.afq_last_error(devs[0].h).decode())
