#!/usr/bin/env python3
"""Per-configuration roofline rows from the rocprofv3 kernel statistics that tools/profile_round.sh collects.

  python tools/roofline_configs.py gpurun_out/<tag> > profiles/rNN_configs_roofline.json

For every BASELINE configuration: throughput (the JSON line tools/bench_configs.py printed under the profiler) and, for
its top kernels, the algorithmic work per launch (flops with 4 per real-by-complex and 8 per complex MAC, or bytes for
the gather kernels: SURVEY 8d conventions), the average duration rocprofv3 measured, and the fraction of the fp64 MFMA
peak (78.6 TFLOP/s) or of the HBM peak (8 TB/s).  With the FETCH_SIZE / WRITE_SIZE passes present, the HBM-side
traffic per launch (FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64 bytes) is added."""
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


def work(name, c):
    """(bound, algorithmic work per launch, unit, note) of a kernel, or None."""
    M, na, nb, K, nw = c['M'], c['na'], c['nb'], c['K'], c['nw']
    nt = na + nb
    cx = 2.0 if c.get('cplx') else 1.0
    if 'prop_fused_kernel' in name:
        return 'mfma', 8.0 * M * M * nt * 8 * nw, 'flop', 'B exp(V) B: 2 + 6 products of M x M by M x (na+nb), 8 flops per complex MAC'
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
        return 'mfma', 4.0 * cx * (tri(na * M) + tri(nb * M)) * nw, 'flop', 'exchange energy, quadratic form on the upper triangle of Atil (per determinant)'
    if name.startswith('void exx_kernel'):
        return 'mfma', 4.0 * cx * K * M * (na * na + nb * nb) * nw, 'flop', 'exchange energy, T intermediate (per determinant)'
    if 'TaylorProb' in name:
        return 'mfma', 8.0 * M * M * nt * nw, 'flop', 'one Taylor product V T (both spins)'
    if 'OneBodyProb' in name:
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
    if 'greens_small_kernel<true>' in name:
        return 'hbm', (2.0 * M * nt * 16) * nw, 'B', 'reads phi, writes Ghalf per walker (latency bound: Gauss-Jordan)'
    if name.startswith('gj_big_kernel'):
        return 'valu', 8.0 * 2 * na ** 3 * nw * 2, 'flop', 'register-resident Gauss-Jordan inverse, 8 N^3 per matrix'
    return None


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
        rows = list(csv.DictReader(open(stats)))
        total = sum(float(r['TotalDurationNs']) for r in rows)
        kern = []
        for r in rows[:14]:
            w = work(r['Name'], c)
            avg = float(r['AverageNs'])
            e = {"kernel": r['Name'][:110], "calls": int(r['Calls']), "avg_us": avg / 1e3,
                 "share_of_kernel_time": float(r['TotalDurationNs']) / total}
            if w:
                bound, amount, unit, note = w
                e["bound"] = bound
                e["work_per_launch"] = amount
                e["work_unit"] = unit
                e["note"] = note
                if unit == 'flop':
                    e["achieved"] = amount / (avg * 1e-9) / 1e12
                    e["peak"] = PEAK_TF
                    e["unit"] = "TFLOP/s"
                    # complex-by-complex products by the 3-multiplication form issue 6 flops per complex MAC where the
                    # algorithmic count says 8: the share of the MFMA pipe actually used is 3/4 of the fraction below
                    if re.search(r', true, \d>\(', r['Name']) or r['Name'].startswith('prop_fused_kernel'):
                        e["issued_fraction_of_peak"] = 0.75 * e["achieved"] / PEAK_TF
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
        out.append({"config": cfg, "sizes": c, "bench_line": line,
                    "roofline": None if dom is None else {"bound": dom["bound"], "kernel": dom["kernel"], "achieved": dom["achieved"],
                                                          "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"],
                                                          "traffic": dom.get("traffic_bytes_per_launch")},
                    "kernels": kern})
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
