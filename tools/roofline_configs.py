#!/usr/bin/env python3
"""Per-configuration roofline rows from the rocprofv3 kernel statistics that tools/profile_round.sh collects.

  python tools/roofline_configs.py gpurun_out/<tag> > profiles/rNN_configs_roofline.json

For every BASELINE configuration: throughput (the JSON line bench.py --config printed under the profiler) and, for
its top kernels, the algorithmic work per launch (flops with 4 per real-by-complex and 8 per complex MAC, or bytes for
the gather kernels: SURVEY 8d conventions), the average duration rocprofv3 measured, and two fractions of the fp64 MFMA
peak (78.6 TFLOP/s):
  frac_algorithmic  the SURVEY 8d count (4 multiplications per complex product, no padding) over the time -- comparable
                    across implementations, can exceed the share of the pipe that is busy (a 3-multiplication kernel
                    tops out at 1.33) and is therefore NOT a utilisation;
  frac_issued       MFMA instructions x their flop count over the time (tile and contraction padding included, a
                    3-multiplication product counted as 3, a real operand as 2), from the launch geometry encoded in the
                    kernel's template arguments -- the utilisation of the matrix pipe, never above 1.
`frac` of a row is frac_issued when it is known (else frac_algorithmic clipped to the ceiling of its product form).
HBM-bound kernels: algorithmic bytes over the time against 8 TB/s.  Per-kernel averages come from the kernel TRACE when it
is present, with launches shorter than 10 % of the median dropped (a population in which every fourth launch is a
near-no-op on dead walkers would otherwise halve the average); else from the statistics file.  With the FETCH_SIZE /
WRITE_SIZE passes present, the HBM-side traffic per launch (FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64
bytes) is added."""
import csv
import json
import os
import re
import sys

PEAK_TF, PEAK_TBS = 78.6, 8.0

CONFIGS = {
    'C1': dict(M=16, na=8, nb=8, K=16, nw=10, kind='hubbard'),
    'C2': dict(M=93, na=7, nb=7, K=1500, nw=256, kind='ueg', nq=750),
    'C4': dict(M=256, na=128, nb=128, K=256, nw=256, kind='hubbard'),
    'C5sd': dict(M=400, na=50, nb=50, K=2000, nw=256, kind='generic', ndet=1, cplx=False),
    'C5': dict(M=400, na=50, nb=50, K=2000, nw=256, kind='generic', ndet=4, cplx=True),
    'bench_c3': dict(M=100, na=25, nb=25, K=500, nw=256, kind='generic', ndet=1, cplx=False),
}


def work(name, c, cs=None):
    """(bound, algorithmic work per launch, unit, note) of a kernel, or None.  cs: the `closed_shell` object of the bench line the
    same run printed (what the device counted: the share of the walker steps that went through a closed-shell deal of the
    propagator / of the large-system GEMM chain, and whether the exchange energy evaluated one spin) -- executed work is priced."""
    M, na, nb, K, nw = c['M'], c['na'], c['nb'], c['K'], c['nw']
    nt = na + nb
    cx = 2.0 if c.get('cplx') else 1.0
    cs = cs or {}
    s_prop, s_chain = float(cs.get('propagator_walker_step_share') or 0.0), float(cs.get('large_chain_walker_step_share') or 0.0)
    closed_note = lambda s: '; closed-shell walkers (%.0f %% of the walker steps): alpha columns only' % (100.0 * s) if s > 0 else ''
    if 'prop_fused_kernel' in name:
        return ('mfma', 8.0 * M * M * (6 * (nt - nb * s_prop) + 2 * nt) * nw, 'flop',
                'B exp(V) B: 2 + 6 products of M x M by M x (na+nb), 8 flops per complex MAC' + closed_note(s_prop).replace('alpha columns only', 'Taylor products on the alpha columns only'))
    if 'VhsProb' in name:
        cols = M * (M + 1) // 2 if c['kind'] == 'generic' else M * M
        return 'mfma', 4.0 * cols * K * nw, 'flop', 'HS potential, packed symmetric columns'
    if 'ForceBiasProb<false>' in name:
        # real-B engine: a complex rchol runs it twice (Re, Im), each launch a real-by-complex product; a single
        # real RHF-type determinant contracts Ghalf_a + Ghalf_b once (both spins share the rchol block)
        rows = na if (c.get('ndet', 1) == 1 and not c.get('cplx') and na == nb) else nt
        return 'mfma', 4.0 * K * rows * M * nw, 'flop', 'force bias / Coulomb vectors (one real-B pass; spin-summed Ghalf when both spins share rchol)'
    if 'ForceBiasProb' in name:
        return 'mfma', 4.0 * cx * K * nt * M * nw, 'flop', 'force bias / Coulomb vectors'
    if 'ExxQProb' in name:
        tri = lambda n: n * (n + 1) / 2.0
        if cs.get('exchange_one_spin'):
            return 'mfma', 4.0 * cx * tri(na * M) * nw, 'flop', 'exchange energy, quadratic form on the upper triangle of Atil: closed-shell population, one spin evaluated and counted twice (the beta launch returns at once)'
        return 'mfma', 4.0 * cx * (tri(na * M) + tri(nb * M)) * nw, 'flop', 'exchange energy, quadratic form on the upper triangle of Atil (per determinant)'
    if name.startswith('void exx_kernel'):
        return 'mfma', 4.0 * cx * K * M * (na * na + nb * nb) * nw, 'flop', 'exchange energy, T intermediate (per determinant)'
    if 'TaylorProb' in name:
        return 'mfma', 8.0 * M * M * (nt - nb * s_chain) * nw, 'flop', 'one Taylor product V T (both spins)' + closed_note(s_chain)
    if 'OneBodyProb' in name:
        # (one spin: doubled below for a launch that holds both -- the closed-shell walkers of such a launch execute alpha's)
        return 'mfma', 8.0 * M * M * na * nw, 'flop', 'BH1 phi, one spin'
    if 'OvlpProb' in name:
        return 'mfma', 8.0 * na * na * M * nw * 2, 'flop', 'phi^T conj(psi), both spins'
    if 'GhalfProb' in name:
        return 'mfma', 8.0 * na * na * M * nw * 2, 'flop', 'O^-1 phi^T, both spins'
    if 'FullGProb' in name:
        return 'mfma', 8.0 * M * M * na * nw, 'flop', 'G = conj(psi) Ghalf, one spin'
    if 'GramProb' in name or 'QProb' in name:
        return 'mfma', 8.0 * na * na * M * nw * 2, 'flop', 'Cholesky-QR GEMM, both spins'
    if name.startswith('vbias_ueg'):
        return 'hbm', (2.0 * M * M * 16 + K * 16) * nw, 'B', 'reads G [2, M, M], writes vbias [K] per walker'
    if name.startswith('vhs_ueg'):
        return 'hbm', (K * 16 + M * M * 16) * nw, 'B', 'reads xs [K], writes VHS [M, M] per walker'
    if 'energy_ueg' in name:
        return 'hbm', (2.0 * M * M * 16) * nw, 'B', 'reads G [2, M, M] per walker'
    if 'greens_small_kernel<true' in name:
        return 'hbm', (2.0 * M * nt * 16) * nw, 'B', 'reads phi, writes Ghalf per walker (latency bound: Gauss-Jordan)'
    if name.startswith('gj_big_kernel'):
        return 'valu', 8.0 * na ** 3 * nw * 2, 'flop', 'register-resident Gauss-Jordan inverse, 8 N^3 flops per N x N complex matrix, 2 nw matrices'
    if 'prop_ueg_kernel' in name:
        return 'mfma', 8.0 * M * M * nt * 6 * nw, 'flop', 'exp(V) phi from per-walker coefficients: 6 products of M x M by M x (na+nb); B is diagonal (row scaling)'
    if 'ueg_fields_kernel' in name:
        return 'hbm', (nt * M * 16 + 2 * K * 16 + 600 * 16) * nw, 'B', 'reads Ghalf, writes xbar, xs and the HS coefficients per walker (latency bound: RNG chain, dependent gathers)'
    return None


def gemm_issued(name, c, merged=False, cs=None):
    """Matrix-pipe flops of one launch of the ring GEMM engine from its template arguments
    mfma_gemm_wg_kernel<WM, WN, TM, TN, D, Prob, MAP, K3M, KC, STAG> and the problem shape; None when unknown."""
    m = re.match(r'void mfma_gemm_wg_kernel<(\d+), (\d+), (\d+), (\d+), \d+, (\w+)(<(\w+)>)?, \d+, (true|false)', name)
    if not m:
        return None
    WM, WN, TM, TN = (int(m.group(i)) for i in range(1, 5))
    prob, targ, k3m = m.group(5), m.group(7), m.group(8) == 'true'
    M, na, nb, K, nw = c['M'], c['na'], c['nb'], c['K'], c['nw']
    nt, nmax = na + nb, max(na, nb)
    tri = False
    cs = cs or {}
    s_chain = float(cs.get('large_chain_walker_step_share') or 0.0)
    closed_cols = None          # columns whose tiles a closed-shell walker of the large-system chain executes
    if prob == 'VhsProb':
        batch, rows, cols, kdim, mults = 1, nw, (M * (M + 1) // 2 if c['kind'] == 'generic' else M * M), K, 2
    elif prob == 'ForceBiasProb':
        contr = (na if (c.get('ndet', 1) == 1 and not c.get('cplx') and na == nb) else nt) * M
        batch, rows, cols, kdim, mults = 1, nw, K, contr, 2
    elif prob == 'ExxQProb':
        batch, rows, cols, kdim, mults, tri = (1 if cs.get('exchange_one_spin') else 2), nw, na * M, na * M, (3 if targ == 'true' else 2), True
    elif prob == 'TaylorProb':
        batch, rows, cols, kdim, mults = nw, M, nt, M, 3 if k3m else 4
        closed_cols = na
    elif prob in ('OneBodyProb', 'OneBodyProbT'):
        batch, rows, cols, kdim, mults = nw, M, (nt if merged else na), M, (2 if targ == 'true' else 3 if k3m else 4)
        closed_cols = na if merged else None
    elif prob in ('OvlpProb', 'OvlpProbT', 'GramProb'):
        batch, rows, cols, kdim, mults = 2 * nw, nmax, nmax, M, (2 if targ == 'true' else 3 if k3m else 4)
    elif prob in ('GhalfProb', 'GhalfProbT'):
        batch, rows, cols, kdim, mults = 2 * nw, nmax, M, nmax, 3 if k3m else 4
    elif prob == 'QProb':
        batch, rows, cols, kdim, mults = 2 * nw, M, nmax, nmax, 3 if k3m else 4
    else:
        return None
    tr, tc = 16 * WM * TM, 16 * WN * TN
    tiles_m, tiles_n = -(-rows // tr), -(-cols // tc)
    if tri:     # triangular operand: column tile tn contracts min(cols, tc (tn + 1)) rows
        ksum = sum(-(-min(kdim, tc * (tn + 1)) // 8) * 8 for tn in range(tiles_n))
        return 2.0 * mults * batch * tiles_m * tr * tc * ksum
    kpad = -(-kdim // 8) * 8
    if closed_cols is not None and s_chain > 0.0:
        # a tile is left out when it lies wholly in the beta columns: the tiles that START below na run
        tiles_n = (1.0 - s_chain) * tiles_n + s_chain * -(-closed_cols // tc)
    return 2.0 * mults * batch * tiles_m * tiles_n * tr * tc * kpad


def issued_flops(name, c, merged=False, cs=None):
    M, na, nb, nw = c['M'], c['na'], c['nb'], c['nw']
    cs = cs or {}
    per = cs.get('propagator_issued_flops_per_walker')
    if 'prop_fused_kernel' in name and per:
        # the library's own count for either deal (afq_propagator_issued_flops), mixed by the device's count of the walker steps
        s = float(cs.get('propagator_walker_step_share') or 0.0)
        return ((1.0 - s) * per['open'] + s * per['closed']) * nw
    if 'prop_fused_kernel' in name:
        nrt, ct, nch = -(-M // 16), -(-na // 16) + -(-nb // 16), -(-M // 8)
        rem4 = 96 < M <= 100
        per_pass = 2.0 * nch * (2048.0 * (nrt - 1 if rem4 else nrt) * ct + 512.0 * (ct if rem4 else 0))
        return (3.0 * 6 + 2.0 * 2.0) * per_pass * nw          # real BH1 (two real products per one-body application)
    if 'prop_ueg_kernel' in name:
        mp = -(-M // 16) * 16
        return 3.0 * 6 * 2.0 * (mp // 16) * (mp // 8) * 2048.0 * nw
    return gemm_issued(name, c, merged, cs)


def trace_averages(path):
    """{kernel name: (average ns over the launches kept, launches kept, launches dropped)} from a rocprofv3 kernel trace."""
    if not os.path.exists(path):
        return {}
    per = {}
    for r in csv.DictReader(open(path)):
        try:
            per.setdefault(r['Kernel_Name'], []).append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        except (KeyError, ValueError):
            return {}
    out = {}
    for k, v in per.items():
        v.sort()
        med = v[len(v) // 2]
        keep = [x for x in v if x >= 0.1 * med]
        out[k] = (sum(keep) / len(keep), len(keep), len(v) - len(keep))
    return out


def main():
    root = sys.argv[1]
    out = []
    for cfg, c in CONFIGS.items():
        name = 'bench_c3' if cfg == 'bench_c3' else 'cfg_' + cfg
        stats = os.path.join(root, name + '_kernel_stats.csv')
        if not os.path.exists(stats):
            continue
        line = None
        try:
            with open(os.path.join(root, name + '.json')) as f:
                for ln in f:
                    if ln.startswith('{'):
                        line = json.loads(ln)
        except OSError:
            pass
        pmc = {}
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            p = os.path.join(root, '%s_pmc_%s.txt' % (name, ctr))
            if os.path.exists(p):
                for ln in open(p):
                    m = re.match(r'(.*\S)\s+%s\s+n=\s*\d+ mean=(\S+)' % ctr, ln)
                    if m:
                        pmc.setdefault(m.group(1)[:60], {})[ctr] = float(m.group(2))
        trace = trace_averages(os.path.join(root, name + '_kernel_trace.csv'))
        rows = list(csv.DictReader(open(stats)))
        total = sum(float(r['TotalDurationNs']) for r in rows)
        kern = []
        # one-body products: the library launches both spins at once when they share one real matrix (k_onebody: two launches
        # per step instead of four) -- the template arguments do not show it, the launch count per step does
        nsteps = next((int(r['Calls']) for r in rows if 'fields_kernel' in r['Name']), 0)
        for r in rows[:14]:
            cs = (line or {}).get('closed_shell')
            w = work(r['Name'], c, cs)
            avg = float(r['AverageNs'])
            merged = 'OneBodyProb' in r['Name'] and nsteps > 0 and int(r['Calls']) / float(nsteps) < 3.0
            e = {"kernel": r['Name'][:110], "calls": int(r['Calls']), "avg_us": avg / 1e3,
                 "share_of_kernel_time": float(r['TotalDurationNs']) / total}
            if r['Name'] in trace:
                avg, kept, dropped = trace[r['Name']]
                e["avg_us"] = avg / 1e3
                e["averaged_over"] = "%d launches of the trace (%d shorter than 10 %% of the median dropped)" % (kept, dropped)
            if w:
                bound, amount, unit, note = w
                if merged:
                    s_chain = float((cs or {}).get('large_chain_walker_step_share') or 0.0)
                    amount, note = (2.0 - s_chain) * amount, 'BH1 phi, both spins in one launch' + (
                        '; closed-shell walkers (%.0f %% of the walker steps): alpha columns only' % (100.0 * s_chain) if s_chain > 0 else '')
                e["bound"] = bound
                e["work_per_launch"] = amount
                e["work_unit"] = unit
                e["note"] = note
                if unit == 'flop':
                    e["achieved"] = amount / (avg * 1e-9) / 1e12
                    e["peak"] = PEAK_TF
                    e["unit"] = "TFLOP/s"
                    e["frac_algorithmic"] = e["achieved"] / PEAK_TF
                    iss = issued_flops(r['Name'], c, merged, cs) if bound == 'mfma' else None
                    if iss:
                        e["issued_flops_per_launch"] = iss
                        e["frac_issued"] = iss / (avg * 1e-9) / 1e12 / PEAK_TF
                        e["frac"] = e["frac_issued"]
                    else:
                        e["frac"] = min(e["frac_algorithmic"], 1.0)
                else:
                    e["achieved"] = amount / (avg * 1e-9) / 1e12
                    e["peak"] = PEAK_TBS
                    e["unit"] = "TB/s (algorithmic bytes)"
                    e["frac"] = e["achieved"] / e["peak"]
            t = pmc.get(r['Name'][:60])
            if t and 'FETCH_SIZE' in t and 'WRITE_SIZE' in t:
                e["traffic_bytes_per_launch"] = (2.0 * t['FETCH_SIZE'] + t['WRITE_SIZE']) * 1024.0
                e["traffic_TBs"] = e["traffic_bytes_per_launch"] / (avg * 1e-9) / 1e12
            kern.append(e)
        dom = max((k for k in kern if 'frac' in k), key=lambda k: k['share_of_kernel_time'], default=None)
        if line and isinstance(line.get('roofline'), dict) and line['roofline'].get('kernel'):
            # bench.py names its dominant kernel itself (per step; the profile also holds the clock pre-heat's energy evaluations)
            first = line['roofline']['kernel'].split(' ')[0].split('<')[0]
            named = [k for k in kern if 'frac' in k and first in k['kernel']]
            if named:
                dom = named[0]
        out.append({"config": cfg, "sizes": c, "bench_line": line,
                    "roofline": None if dom is None else {"bound": dom["bound"], "kernel": dom["kernel"], "achieved": dom["achieved"],
                                                          "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"],
                                                          "frac_algorithmic": dom.get("frac_algorithmic"),
                                                          "frac_issued": dom.get("frac_issued"),
                                                          "traffic": dom.get("traffic_bytes_per_launch")},
                    "kernels": kern})
    bad = [(o["config"], k["kernel"][:50], k["frac"]) for o in out for k in o["kernels"] if k.get("frac", 0) > 1.0]
    if bad:
        sys.stderr.write("roofline_configs: fractions above 1: %r\n" % (bad,))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
