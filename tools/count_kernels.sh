#!/bin/bash
# usage: count_kernels.sh lib.so  -> number of kernels (.kd symbols) in the gfx950 code objects of the fat binary
so=$1
python3 - "$so" <<'PY'
import sys, subprocess, re, os, tempfile
data = open(sys.argv[1], 'rb').read()
# clang offload bundle(s): magic "__CLANG_OFFLOAD_BUNDLE__"
magic = b"__CLANG_OFFLOAD_BUNDLE__"
n = 0; total = 0; pos = 0
import struct
while True:
    i = data.find(magic, pos)
    if i < 0: break
    nb = struct.unpack_from('<Q', data, i + 24)[0]
    off = i + 32
    for _ in range(nb):
        eo, es, ts = struct.unpack_from('<QQQ', data, off)
        triple = data[off + 24: off + 24 + ts].decode()
        off += 24 + ts
        if 'gfx950' in triple and es:
            with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
                f.write(data[i + eo: i + eo + es]); name = f.name
            out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '-s', '-W', name], capture_output=True, text=True).stdout
            k = len(set(l.split()[-1] for l in out.splitlines() if l.rstrip().endswith('.kd')))   # (.dynsym and .symtab list each)
            n += k; total += es
            os.unlink(name)
    pos = i + 24
print("kernels: %d   device code bytes: %d" % (n, total))
PY
