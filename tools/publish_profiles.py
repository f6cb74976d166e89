#!/usr/bin/env python3
"""Copy the evidence of one tools/profile_round.sh run into profiles/ under the round's names and derive the two summaries
bench.py and the documents read:  python tools/publish_profiles.py gpurun_out/<tag> r03 [standalone bench line file]
  profiles/<round>_<name>_kernel_stats.csv, _pmc_FETCH_SIZE.txt, _pmc_WRITE_SIZE.txt      copies
  profiles/<round>_bench_c3_under_rocprof.json                                             the bench line of the profiled run
  profiles/<round>_bench_c3.json                                                           the bench line of a run without profiler
  profiles/<round>_configs_roofline.json                                                   tools/roofline_configs.py
  profiles/<round>_pmc_traffic.json        HBM-side bytes per launch of every bench kernel: (2 FETCH_SIZE + WRITE_SIZE) KB, the
                                           gfx950 correction of MI355X_MICROARCH.md, next to the kernel's duration in that run"""
import csv
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json_line(path):
    line = None
    with open(path) as f:
        for ln in f:
            if ln.startswith('{'):
                line = ln
    return line


def short(name):
    m = re.match(r'(?:void )?(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)(<.*)?', name)
    base = m.group(1)
    if base == 'mfma_gemm_wg_kernel' or base == 'mfma_gemm_kernel':
        p = re.search(r'(\w+Prob)', name)
        return '%s<%s>' % (base, p.group(1)) if p else base
    return base


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, 'profiles')
    for f in sorted(os.listdir(src)):
        if f.endswith('_kernel_stats.csv') or re.search(r'_pmc_(FETCH|WRITE)_SIZE\.txt$', f):
            shutil.copy(os.path.join(src, f), os.path.join(prof, '%s_%s' % (rnd, f)))
    ln = last_json_line(os.path.join(src, 'bench_c3.json'))
    if ln:
        open(os.path.join(prof, rnd + '_bench_c3_under_rocprof.json'), 'w').write(ln)
    if len(sys.argv) > 3:
        ln = last_json_line(sys.argv[3])
        if ln:
            open(os.path.join(prof, rnd + '_bench_c3.json'), 'w').write(ln)
    with open(os.path.join(prof, rnd + '_configs_roofline.json'), 'w') as out:
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'roofline_configs.py'), src], stdout=out)
    # ---- traffic per launch of the bench kernels, and of the kernels of the other configurations
    def traffic_of(name):
        pmc = {}
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            path = os.path.join(src, '%s_pmc_%s.txt' % (name, ctr))
            if not os.path.exists(path):
                return {}
            for ln in open(path):
                m = re.match(r'(.*\S)\s+%s\s+n=\s*\d+ mean=(\S+)' % ctr, ln)
                if m:
                    pmc.setdefault(m.group(1)[:60], {})[ctr] = float(m.group(2))
        out = {}
        stats = os.path.join(src, '%s_kernel_stats.csv' % name)
        if not os.path.exists(stats):
            return {}
        for r in csv.DictReader(open(stats)):
            t = pmc.get(r['Name'][:60])
            if t and 'FETCH_SIZE' in t and 'WRITE_SIZE' in t and not r['Name'].startswith('__amd'):
                out.setdefault(short(r['Name']), {
                    "traffic_bytes_per_launch": (2.0 * t['FETCH_SIZE'] + t['WRITE_SIZE']) * 1024.0,
                    "avg_us_in_that_run": float(r['AverageNs']) / 1e3})
        return out
    tag = os.path.basename(os.path.normpath(src))
    traffic = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 20 --warmup 5 "
                  "--no-cpu-baseline, and -- python3 bench.py --config C2 | C4 | C5 --no-cpu-baseline --repeats 1 under "
                  "'configs' (tools/profile_round.sh %s); per-kernel means in profiles/%s_*_pmc_FETCH_SIZE.txt / "
                  "_WRITE_SIZE.txt" % (tag, rnd),
        "note": "FETCH_SIZE / WRITE_SIZE are KB of L2 memory-side requests (Infinity-Cache hits included); FETCH_SIZE is doubled "
                "on gfx950 as MI355X_MICROARCH.md prescribes.  One name = one kernel template: a mean over every launch of "
                "that template in the run (the per-determinant and the averaged-G force-bias contractions of C5 share one)"}
    sys.path.insert(0, ROOT)
    import bench
    traffic["csrc_sha16"] = bench.csrc_sha16()          # the library the passes ran on (the snapshot they were made from)
    traffic.update(traffic_of('bench_c3'))
    traffic["configs"] = {c: traffic_of('cfg_' + c) for c in ('C2', 'C4', 'C5') if traffic_of('cfg_' + c)}
    json.dump(traffic, open(os.path.join(prof, rnd + '_pmc_traffic.json'), 'w'), indent=1)
    print("published", src, "->", prof, "as", rnd)


if __name__ == "__main__":
    main()
