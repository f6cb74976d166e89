"""Pin the CPU oracle (oracle/afqmc_ref.py) against vectors produced by the
genuine reference (tests/golden/make_golden.py) and against the known-answer
constants of the reference's own tests.  CPU only."""
import numpy
import pytest

from oracle import afqmc_ref as ref
from tests.helpers import generic_model, hubbard_model, msd_model, ueg_model

RTOL = 1e-11


def close(a, b, tol=RTOL):
    a = numpy.asarray(a)
    b = numpy.asarray(b)
    scale = max(1.0, float(numpy.max(numpy.abs(b)))) if b.size else 1.0
    assert numpy.max(numpy.abs(a - b)) <= tol * scale, numpy.max(numpy.abs(a - b)) / scale


def check_single_walker_ops(d, tag, model):
    na, nb, M = model.na, model.nb, model.M
    phi = d[tag + 'phi']
    det, Ghalf, G = ref.greens_function(phi, model.psi, na, nb)
    close(det, d[tag + 'det'])
    close(Ghalf[0], d[tag + 'Ghalf_a'])
    close(Ghalf[1], d[tag + 'Ghalf_b'])
    close(G, d[tag + 'G'])
    if tag + 'ovlp' in d:
        close(ref.calc_overlap(phi, model.psi, na, nb), d[tag + 'ovlp'])
        # same determinant as the Green's function overlap
        close(ref.calc_overlap(phi, model.psi, na, nb), det)
    xbar = model.force_bias(Ghalf, G)
    close(xbar, d[tag + 'xbar'])
    xi = d[tag + 'xi']
    xs, cmf, cfb, ntrig = ref.shift_fields(xi, d[tag + 'xbar_big'], model.mf_shift, model.sqrt_dt)
    assert ntrig >= 1
    close(xs, d[tag + 'xs_clip'])
    close(cmf, d[tag + 'cmf_clip'])
    close(cfb, d[tag + 'cfb_clip'])
    xs, cmf, cfb, ntrig = ref.shift_fields(xi, xbar, model.mf_shift, model.sqrt_dt)
    VHS = model.vhs(xs)
    close(VHS, d[tag + 'VHS'])
    p2 = phi.copy()
    if VHS.ndim == 3:
        ref.apply_exponential(p2[:, :na], VHS[0])
        ref.apply_exponential(p2[:, na:], VHS[1])
    else:
        ref.apply_exponential(p2[:, :na], VHS)
        ref.apply_exponential(p2[:, na:], VHS)
    close(p2, d[tag + 'phi_exp'])
    p3 = phi.copy()
    ref.kinetic_real(p3, model.BH1, na)
    close(p3, d[tag + 'phi_kin'])
    close(numpy.array(model.local_energy(G, Ghalf)), d[tag + 'energy'])
    if tag + 'step_phi' in d:
        w = ref.new_walker(model, phi)
        w['hybrid_energy'] = 0.25 + 0.1j
        ref.propagate_walker_phaseless(model, w, xi, 0.3)
        close(w['phi'], d[tag + 'step_phi'])
        close(w['weight'], d[tag + 'step_weight'])
        close(w['ot'], d[tag + 'step_ot'])
        close(w['hybrid_energy'], d[tag + 'step_ehyb'])
    p4 = phi.copy()
    detR = ref.reortho(p4, na, nb)
    close(p4, d[tag + 'phi_qr'])
    close(detR, d[tag + 'detR'])


def test_generic_ops(golden):
    d = golden('generic_ops.npz')
    for tag in ('A_', 'B_'):
        check_single_walker_ops(d, tag, generic_model(d, tag))


def test_generic_energy_known_answers(golden):
    """estimators/tests/test_generic.py:45-47,62-64: half-rotated == full-G energy."""
    d = golden('generic_ops.npz')
    m = generic_model(d, 'A_')
    pinned = (20.6826247016273, 23.0173528796140, -2.3347281779866)
    G = d['A_trialG']
    GH = [d['A_trialGH_a'], d['A_trialGH_b']]
    e_opt = ref.local_energy_generic_cholesky_opt(m.H1, m.ecore, G, GH, m.rchol, m.na, m.nb)
    e_full = ref.local_energy_generic_cholesky(m.H1, m.ecore, G, m.hs_pot)
    for e in (e_opt, e_full, d['A_e_opt'], d['A_e_full']):
        assert numpy.array(e).real == pytest.approx(pinned, rel=1e-10)


def test_hubbard_ops(golden):
    d = golden('hubbard_ops.npz')
    check_single_walker_ops(d, 'C_', hubbard_model(d, 'C_', 'hubbard'))
    check_single_walker_ops(d, 'S_', hubbard_model(d, 'S_', 'hubbard_spin'))


def test_hubbard_spin_known_answer(golden):
    """propagation/tests/test_hubbard.py:118-136: walker.ovlp.real == 0.765551499039435."""
    d = golden('hubbard_ops.npz')
    m = hubbard_model(d, 'S_', 'hubbard_spin')
    w = ref.new_walker(m, m.psi)
    ref.propagate_walker_phaseless(m, w, d['pin_xi'], 0.0)
    assert w['ovlp'].real == pytest.approx(0.765551499039435, rel=1e-11)
    assert abs(w['ovlp'].imag) < 1e-12


def test_ueg_ops(golden):
    d = golden('ueg_ops.npz')
    m = ueg_model(d, 'U_')
    check_single_walker_ops(d, 'U_', m)
    # propagation/tests/test_planewave.py:12-38
    _, Ghalf, G = ref.greens_function(d['pw_phi'], m.psi, m.na, m.nb)
    fb = m.force_bias(Ghalf, G)
    assert numpy.linalg.norm(fb) == pytest.approx(0.16660828645573392, rel=1e-11)
    vhs = m.vhs(d['pw_xi'] - fb)
    assert numpy.linalg.norm(vhs) == pytest.approx(0.1467322554815581, rel=1e-11)


def run_traj(d, model, **kw):
    nw = d['phi0'].shape[0]
    walkers = [ref.new_walker(model, d['phi0'][i]) for i in range(nw)]
    xi, r = d['xi'], d['r']

    def xi_source(step, iw):
        x = xi[step - 1, iw]
        assert not numpy.isnan(x[0]), "oracle propagated a walker the reference skipped"
        return x

    rec = []
    blocks = ref.run_afqmc(model, walkers, xi_source, lambda step: r[step - 1],
                           int(d['nsteps']), int(d['nblocks']), nstblz=int(d['nstblz']),
                           npop_control=int(d['npop_control']),
                           energy_eval_freq=int(d['energy_eval_freq']), record=rec, **kw)
    return walkers, rec, blocks


def check_traj(d, model, tol=1e-8, mean_etotal=None, **kw):
    walkers, rec, blocks = run_traj(d, model, **kw)
    fp = kw.get('free_projection', False)
    if 'phase' in d:
        close(numpy.array([x['phase'] for x in rec]), d['phase'], tol)
        close(numpy.array([x['eloc'] for x in rec]), d['eloc'], tol)
    W = numpy.array([x['weight'] for x in rec])
    close(W, d['weight'], tol)
    close(numpy.array([x['unscaled_weight'] for x in rec]), d['unscaled_weight'], tol)
    close(numpy.array([x['ot'] for x in rec]), d['ot'], tol)
    close(numpy.array([x['hybrid_energy'] for x in rec]), d['ehyb'], tol)
    pix = numpy.array([x['parent_ix'] for x in rec if x['parent_ix'] is not None])
    assert numpy.array_equal(pix, d['parent_ix'])
    B = numpy.array(blocks)
    gold = d['blocks']                      # [nblocks+1, 1 + 10]: step, estimates
    close(B[:, :9], gold[:, 1:10], tol)
    if mean_etotal is not None:
        # numpy.mean(extract_mixed_estimates('estimates.0.h5').ETotal.values[:-1]), qmc/tests/test_afqmc.py:188,229
        assert numpy.mean(B[:-1, 4].real) == pytest.approx(mean_etotal, rel=1e-9)
    close(numpy.array([w['phi'] for w in walkers]), d['final_phi'], tol)
    # final estimator pass pinned by the reference's driver tests
    est = numpy.zeros(10, dtype=numpy.complex128)
    ref.mixed_update(model, est, walkers, 0, 1, fp)
    close(est[:9], d['final_estimates'][:9], tol)
    return est


def test_traj_generic(golden):
    d = golden('traj_generic.npz')
    est = check_traj(d, generic_model(d, ''), mean_etotal=1.5485077038208)
    # qmc/tests/test_afqmc.py:227
    assert est[ref.EST['enumer']].real == pytest.approx(3.8763193646854273, rel=1e-9)


def test_traj_hubbard(golden):
    d = golden('traj_hubbard.npz')
    est = check_traj(d, hubbard_model(d, '', 'hubbard'), mean_etotal=-15.14323385684513)
    # qmc/tests/test_afqmc.py:186
    assert est[ref.EST['enumer']].real == pytest.approx(-152.91937839611, rel=1e-9)


def test_traj_hubbard_c1(golden):
    d = golden('traj_hubbard_c1.npz')
    check_traj(d, hubbard_model(d, '', 'hubbard'))
    assert d['parent_ix'].shape[0] == 20           # comb every 5 steps


def test_traj_hubbard_free_projection(golden):
    """propagation/continuous.py:175-200 + estimators/mixed.py:151-175."""
    d = golden('traj_hubbard_fp.npz')
    assert bool(d['free_projection'])
    check_traj(d, hubbard_model(d, '', 'hubbard'), free_projection=True)


def test_traj_hubbard_local_energy_weights(golden):
    """propagation/continuous.py:294-318 (hybrid: False)."""
    d = golden('traj_hubbard_le.npz')
    assert not bool(d['hybrid'])
    check_traj(d, hubbard_model(d, '', 'hubbard'), hybrid=False)


def test_traj_ueg(golden):
    d = golden('traj_ueg.npz')
    est = check_traj(d, ueg_model(d, '', 'sys_'))
    # qmc/tests/test_afqmc.py:87-92
    assert est[ref.EST['enumer']].real == pytest.approx(16.33039729324558, rel=1e-9)
    assert est[ref.EST['uweight']].real == pytest.approx(9.75405059997262, rel=1e-9)


def check_msd_steps(d, tag, hybrid):
    m = msd_model(d, tag)
    phi0 = d[tag + 'phi0']
    tot, weights, Gi = m.greens(phi0)
    close(tot, d[tag + 'ot0'])
    close(weights, d[tag + 'weights0'])
    close(Gi, d[tag + 'Gi0'])
    close(m.overlap(phi0), d[tag + 'ot0'])
    close(m.force_bias(weights, Gi), d[tag + 'xbar0'])
    close(numpy.array(m.local_energy(Gi, weights)), d[tag + 'energy0'])
    w = ref.new_walker(m, phi0)
    eshift = complex(d[tag + 'eshift'])
    for i, xi in enumerate(d[tag + 'xi']):
        ref.propagate_walker_phaseless(m, w, xi, eshift, hybrid=hybrid)
        close(w['phi'], d[tag + 'step_phi'][i], 1e-9)
        close(w['weight'], d[tag + 'step_weight'][i], 1e-9)
        close(w['ot'], d[tag + 'step_ot'][i], 1e-9)
        if hybrid:
            close(w['hybrid_energy'], d[tag + 'step_ehyb'][i], 1e-9)
        else:
            close(w['eloc'], d[tag + 'step_eloc'][i], 1e-9)
    return w


def test_msd_phmsd_known_answers(golden):
    """propagation/tests/test_generic.py:52-92: walker weight after 10 steps with a 3-determinant
    particle-hole trial, local-energy and hybrid weight updates."""
    d = golden('msd_ops.npz')
    w = check_msd_steps(d, 'PL_', False)
    assert w['weight'] == pytest.approx(0.68797524675701, rel=1e-10)
    w = check_msd_steps(d, 'PH_', True)
    assert w['weight'] == pytest.approx(0.7430443466368197, rel=1e-10)


def test_msd_nomsd_ops(golden):
    d = golden('msd_ops.npz')
    assert not bool(d['N_ortho'])
    check_msd_steps(d, 'N_', True)


def test_traj_msd(golden):
    """Unchanged reference driver with a 3-determinant NOMSD trial (8 walkers, comb/5, reortho/5)."""
    d = golden('traj_msd.npz')
    check_traj(d, msd_model(d, ''))
    assert d['parent_ix'].shape[0] == 8


def run_bp(d, restore):
    m = generic_model(d, '')
    nw = d['phi0'].shape[0]
    walkers = [ref.new_walker(m, d['phi0'][i]) for i in range(nw)]
    xi, r = d['xi'], d['r']
    bp = []
    ref.run_afqmc(m, walkers, lambda s, w: xi[s - 1, w], lambda s: r[s - 1], int(d['nsteps']), int(d['nblocks']),
                  nstblz=int(d['nstblz']), npop_control=int(d['npop_control']),
                  energy_eval_freq=int(d['energy_eval_freq']), nbp=int(d['nbp']), bp_out=bp,
                  restore_weights=restore)
    bp = numpy.array(bp)
    close(bp[:, 3], d['bp_denominator'], 1e-9)
    close(bp[:, 4:].reshape(d['bp_one_rdm'].shape), d['bp_one_rdm'], 1e-9)
    return bp[:, 4:].reshape(d['bp_one_rdm'].shape) / bp[:, 3][:, None, None, None]


def test_back_propagated_rdm(golden):
    """qmc/tests/test_afqmc.py:232-278: back-propagated one-body RDM, tau_bp = 5 steps."""
    rdm = run_bp(golden('traj_bp.npz'), None)
    assert rdm[0, 0].trace() == pytest.approx(3.0, rel=1e-10)
    assert rdm[0, 1].trace() == pytest.approx(3.0, rel=1e-10)
    assert rdm[11, 0, 1, 3].real == pytest.approx(-0.121883381144845, rel=1e-9)


def test_back_propagated_rdm_restored_weights(golden):
    """estimators/back_propagation.py:187-199 with restore_weights == "full"."""
    d = golden('traj_bp_full.npz')
    assert str(d['restore_weights']) == 'full'
    run_bp(d, 'full')


def test_back_propagated_rdm_two_path_lengths(golden):
    """estimators/back_propagation.py:68-69,145-147,219-222 with nsplit = 2: windows of 3 and 6 steps from one field
    history, reset only at the longer one."""
    d = golden('traj_bp_split.npz')
    m = generic_model(d, '')
    nw = d['phi0'].shape[0]
    walkers = [ref.new_walker(m, d['phi0'][i]) for i in range(nw)]
    xi, r = d['xi'], d['r']
    bp = []
    ref.run_afqmc(m, walkers, lambda s, w: xi[s - 1, w], lambda s: r[s - 1], int(d['nsteps']), int(d['nblocks']),
                  nstblz=int(d['nstblz']), npop_control=int(d['npop_control']),
                  energy_eval_freq=int(d['energy_eval_freq']), nbp=int(d['nbp']), bp_out=bp, bp_nsplit=2)
    assert [int(x) for x in d['splits']] == [3, 6]
    for sp in (3, 6):
        mine = numpy.array([e for (k, e) in bp if k == sp])
        close(mine[:, 3], d['bp_denominator_%d' % sp], 1e-9)
        close(mine[:, 4:].reshape(d['bp_one_rdm_%d' % sp].shape), d['bp_one_rdm_%d' % sp], 1e-9)


def test_back_propagated_rdm_ueg(golden):
    """estimators/back_propagation.py:127-226 with propagation/planewave.py:114-178 (tau_bp = 4 steps)."""
    d = golden('traj_bp_ueg.npz')
    m = ueg_model(d, '', 'sys_')
    nw = d['phi0'].shape[0]
    walkers = [ref.new_walker(m, d['phi0'][i]) for i in range(nw)]
    xi, r = d['xi'], d['r']
    bp = []
    ref.run_afqmc(m, walkers, lambda s, w: xi[s - 1, w], lambda s: r[s - 1], int(d['nsteps']), int(d['nblocks']),
                  nstblz=int(d['nstblz']), npop_control=int(d['npop_control']),
                  energy_eval_freq=int(d['energy_eval_freq']), nbp=int(d['nbp']), bp_out=bp)
    bp = numpy.array(bp)
    close(bp[:, 3], d['bp_denominator'], 1e-9)
    close(bp[:, 4:].reshape(d['bp_one_rdm'].shape), d['bp_one_rdm'], 1e-9)


def test_mixed_one_rdm(golden):
    """estimators/mixed.py:226-233,279-283 (one_rdm: True, energy every 5 steps: the accumulated walker.G is the
    Green's function before the step's propagation on the steps in between)."""
    d = golden('traj_hubbard_rdm.npz')
    m = hubbard_model(d, '', 'hubbard')
    m.track_G = True
    nw = d['phi0'].shape[0]
    walkers = [ref.new_walker(m, d['phi0'][i]) for i in range(nw)]
    xi, r = d['xi'], d['r']
    rdm = []
    ref.run_afqmc(m, walkers, lambda s, w: xi[s - 1, w], lambda s: r[s - 1], int(d['nsteps']), int(d['nblocks']),
                  nstblz=int(d['nstblz']), npop_control=int(d['npop_control']),
                  energy_eval_freq=int(d['energy_eval_freq']), rdm_out=rdm)
    close(numpy.array(rdm), d['mixed_one_rdm'], 1e-9)


def test_use_log_shift(golden):
    """walkers/handler.py:228,456-475 + single_det.py:159,192,250-253,320 (use_log_shift: True): the weights do not
    see the shift on the continuous path (both overlaps of a step carry the same one), walker.ot and the
    overlap column of the estimates do."""
    d = golden('traj_hubbard_logshift.npz')
    m = hubbard_model(d, '', 'hubbard')
    nw = d['phi0'].shape[0]
    walkers = [ref.new_walker(m, d['phi0'][i]) for i in range(nw)]
    xi, r = d['xi'], d['r']
    rec = []
    blocks = ref.run_afqmc(m, walkers, lambda s, w: xi[s - 1, w], lambda s: r[s - 1], int(d['nsteps']),
                           int(d['nblocks']), nstblz=int(d['nstblz']), npop_control=int(d['npop_control']),
                           energy_eval_freq=int(d['energy_eval_freq']), record=rec, use_log_shift=True)
    close(numpy.array([x['weight'] for x in rec]), d['weight'], 1e-9)
    close(numpy.array([x['ot'] for x in rec]), d['ot'], 1e-9)
    close(numpy.array(blocks)[:, :9], d['blocks'][:, 1:10], 1e-9)
    assert m.log_shift == pytest.approx(d['final_log_shift'][0].real, rel=1e-10)
    assert m.detR_shift == pytest.approx(d['final_detR_shift'][0].real, rel=1e-10)
    assert abs(m.log_shift) > 0.5          # the option does something in this run
    assert numpy.all(d['final_log_shift'] == d['final_log_shift'][0])


def run_hirsch(d, use_log_shift=False, bp_out=None, free_projection=False):
    na, nb = [int(x) for x in d['nelec']]
    m = ref.HirschModel(d['T'], float(d['U']), d['psi'], na, nb, float(d['dt']), bool(d['charge']))
    m.single_site = bool(d['single_site']) if 'single_site' in d else True
    close(m.bt2, d['bt2'])
    nw = d['phi0'].shape[0]
    walkers = [ref.new_walker(m, d['phi0'][i]) for i in range(nw)]
    u, off = d['u'], d['u_off']

    def usrc(step):
        it = iter(u[off[step - 1]:off[step]])
        return lambda: next(it)

    rec = []
    blocks = ref.run_afqmc(m, walkers, None, None, int(d['nsteps']), int(d['nblocks']), nstblz=int(d['nstblz']),
                           npop_control=int(d['npop_control']), energy_eval_freq=int(d['energy_eval_freq']),
                           record=rec, uniform_source=usrc, hybrid=False, use_log_shift=use_log_shift,
                           nbp=int(d['nbp']) if bp_out is not None else None, bp_out=bp_out,
                           free_projection=free_projection)
    if free_projection:
        close(numpy.array([x['phase'] for x in rec]), d['phase'], 1e-9)
    if use_log_shift:
        assert m.log_shift == pytest.approx(d['final_log_shift'][0].real, rel=1e-10)
        assert m.detR_shift == pytest.approx(d['final_detR_shift'][0].real, rel=1e-10)
    close(numpy.array([x['weight'] for x in rec]), d['weight'], 1e-9)
    close(numpy.array([x['ot'] for x in rec]), d['ot'], 1e-9)
    pix = numpy.array([x['parent_ix'] for x in rec if x['parent_ix'] is not None])
    assert numpy.array_equal(pix, d['parent_ix'])
    close(numpy.array(blocks)[:, :9], d['blocks'][:, 1:10], 1e-9)
    close(numpy.array([w['phi'] for w in walkers]), d['final_phi'], 1e-9)
    est = numpy.zeros(10, dtype=numpy.complex128)
    ref.mixed_update(m, est, walkers, 0, 1, free_projection)
    close(est[:9], d['final_estimates'][:9], 1e-9)
    return est, numpy.array(blocks)


def test_traj_hirsch_use_log_shift(golden):
    """Discrete fields with use_log_shift: calc_otrial shifts the determinant of the INVERSE overlap
    (single_det.py:159), so a change of the shift at a population control shows in the next kinetic
    importance-sampling ratio (hubbard.py:163-164) and therefore in the weights."""
    d = golden('traj_hirsch_logshift.npz')
    run_hirsch(d, use_log_shift=True)
    plain = golden('traj_hubbard_hirsch.npz')
    n = d['weight'].shape[0]
    assert numpy.max(numpy.abs(d['unscaled_weight'] - plain['unscaled_weight'][:n])) > 1e-3   # the option changes the weights


def test_back_propagated_rdm_hirsch(golden):
    """estimators/back_propagation.py:127-226 with propagation/hubbard.py:568-600,634-672: fields recorded one site
    at a time (hubbard.py:215-216), tau_bp = 4 steps."""
    d = golden('traj_hirsch_bp.npz')
    bp = []
    run_hirsch(d, bp_out=bp)
    bp = numpy.array(bp)
    close(bp[:, 3], d['bp_denominator'], 1e-9)
    close(bp[:, 4:].reshape(d['bp_one_rdm'].shape), d['bp_one_rdm'], 1e-9)
    rdm = bp[:, 4:].reshape(d['bp_one_rdm'].shape) / bp[:, 3][:, None, None, None]
    assert rdm[0, 0].trace() == pytest.approx(7.0, rel=1e-10)


def test_traj_hubbard_hirsch(golden):
    """qmc/tests/test_afqmc.py:99-143: discrete Hirsch HS with single-site updates."""
    est, blocks = run_hirsch(golden('traj_hubbard_hirsch.npz'))
    assert est[ref.EST['enumer']].real == pytest.approx(-152.68468568462666, rel=1e-10)


def test_traj_hubbard_hirsch_charge(golden):
    d = golden('traj_hubbard_hirsch_charge.npz')
    assert bool(d['charge'])
    run_hirsch(d)


def test_traj_hirsch_direct_update(golden):
    """propagation/hubbard.py:222-275 (two_body_direct, ``single_site_update: False``) against trajectories of the
    reference itself, spin and charge decomposition (tests/golden/make_golden.py hirsch_direct)."""
    for name in ('traj_hirsch_direct.npz', 'traj_hirsch_direct_charge.npz'):
        d = golden(name)
        assert not bool(d['single_site'])
        run_hirsch(d)


def test_traj_hirsch_free_projection(golden):
    """propagation/hubbard.py:303-343 (propagate_walker_free) through the reference driver: spin decomposition
    (aux_wfac = 1: the weights only move with the comb and the shift) and charge decomposition (complex aux_wfac:
    the phase rotates and |wfac| enters the weight)."""
    d = golden('traj_hirsch_fp.npz')
    assert bool(d['free_projection']) and not bool(d['charge'])
    run_hirsch(d, free_projection=True)
    d = golden('traj_hirsch_fp_charge.npz')
    assert bool(d['free_projection']) and bool(d['charge'])
    run_hirsch(d, free_projection=True)
    assert numpy.max(numpy.abs(d['phase'] - 1.0)) > 1e-3            # the phase does rotate in this run


def test_comb_truncation_quirk():
    """walkers/handler.py:301: zip(clone, kill) copies a multiplicity-3 parent once."""
    w = numpy.array([3.0, 1e-9, 1e-9, 1.0 - 2e-9])
    pix = ref.comb_parent_ix(w / (w.sum() / 4), 4, 0.5)
    assert list(pix) == [3, 0, 0, 1]
    assert ref.comb_pairs(pix) == [(0, 1)]


@pytest.mark.parametrize("name", ['traj_generic_c3.npz', 'traj_generic_c3_open.npz'])
def test_traj_generic_at_the_benchmark_size(golden, name):
    """BASELINE configs[2] at its own size and cadence (M = 100, K = 500, 25 + 25 electrons, RHF trial, re-orthogonalisation
    every 10 steps, comb every 5, energy every 10; 36 walkers -- above the 32 from which the device takes its spin-summed / one-spin paths -- 30 steps, four forced comb events that clone and kill) through the genuine
    driver (qmc/afqmc.py:223-255, estimators/mixed.py:133-289, walkers/handler.py:225-338): the oracle against it.  The
    second fixture starts every third walker with a perturbed beta block (open-shell walkers beside closed ones)."""
    from tests import c3_traj
    d = golden(name)
    system, trial, BH1, mf_shift = c3_traj.inputs(d)
    na, nb = system.nup, system.ndown
    model = ref.RefModel('generic', system.nbasis, na, nb, trial.psi, BH1, mf_shift, float(d['dt']), hs_pot=system.hs_pot,
                         rchol=trial._rchol, H1=system.H1.astype(complex), ecore=system.ecore)
    phi0 = c3_traj.initial_walkers(d, trial)
    close(phi0.sum(axis=(1, 2)), d['phi0_sum'], 1e-10)
    walkers = [ref.new_walker(model, p) for p in phi0]
    rp = c3_traj.StateReplay(d)
    rec = []
    blocks = ref.run_afqmc(model, walkers, lambda step, iw: rp.normal(0.0, 1.0, model.nfields), lambda step: rp.random(),
                           int(d['nsteps']), int(d['nblocks']), nstblz=int(d['nstblz']),
                           npop_control=int(d['npop_control']), energy_eval_freq=int(d['energy_eval_freq']), record=rec)
    tol = 1e-8
    for key, gold in (('weight', 'weight'), ('unscaled_weight', 'unscaled_weight'), ('ot', 'ot'), ('hybrid_energy', 'ehyb'),
                      ('phase', 'phase'), ('eloc', 'eloc')):
        close(numpy.array([x[key] for x in rec]), d[gold], tol)
    pix = numpy.array([x['parent_ix'] for x in rec if x['parent_ix'] is not None])
    assert numpy.array_equal(pix, d['parent_ix'])
    assert (pix != 1).any()                                     # walkers were cloned and killed
    close(numpy.array(blocks)[:, :9], d['blocks'][:, 1:10], tol)
    fp = numpy.array([w['phi'] for w in walkers])
    close(numpy.linalg.norm(fp, axis=1), d['final_phi_colnorm'], tol)
    close(fp.sum(axis=1), d['final_phi_sum'], tol)
    est = numpy.zeros(10, dtype=numpy.complex128)
    ref.mixed_update(model, est, walkers, 0, 1, False)
    close(est[:9], d['final_estimates'][:9], tol)
