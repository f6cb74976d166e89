#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every gfx950 kernel in an object / shared library (clang offload bundles inside):
   python tools/kernel_regs.py pauxy_amd/csrc/k_fused.o [name filter]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    data = open(path, 'rb').read()
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return
        nb = struct.unpack_from('<Q', data, i + 24)[0]
        off = i + 32
        for _ in range(nb):
            eo, es, ts = struct.unpack_from('<QQQ', data, off)
            triple = data[off + 24: off + 24 + ts].decode()
            off += 24 + ts
            if 'gfx950' in triple and es:
                yield data[i + eo: i + eo + es]
        pos = i + 24


def main():
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    for co in code_objects(sys.argv[1]):
        with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
            f.write(co)
        out = subprocess.run([READELF, '--notes', f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        for blk in out.split('- .agpr_count:')[1:]:
            g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
            name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
            if flt in name:
                print("vgpr %3s agpr %3s sgpr %3s spill v/s %s/%s scratch %s lds %s  %s" % (
                    g('vgpr_count'), blk.split()[0], g('sgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'),
                    g('private_segment_fixed_size'), g('group_segment_fixed_size'), name[:150]))


if __name__ == '__main__':
    main()
