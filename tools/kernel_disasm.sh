#!/bin/bash
# usage: tools/kernel_disasm.sh obj.o out.s  -- disassembly of the gfx950 code object(s) inside an object / shared library
python3 - "$1" "$2" <<'PY'
import sys, struct, subprocess, tempfile, os
data = open(sys.argv[1], 'rb').read(); magic = b"__CLANG_OFFLOAD_BUNDLE__"; pos = 0; out = open(sys.argv[2], 'w')
while True:
    i = data.find(magic, pos)
    if i < 0: break
    nb = struct.unpack_from('<Q', data, i + 24)[0]; off = i + 32
    for _ in range(nb):
        eo, es, ts = struct.unpack_from('<QQQ', data, off); triple = data[off + 24: off + 24 + ts].decode(); off += 24 + ts
        if 'gfx950' in triple and es:
            with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f: f.write(data[i + eo: i + eo + es])
            out.write(subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '-d', '--no-show-raw-insn', f.name], capture_output=True, text=True).stdout)
            os.unlink(f.name)
    pos = i + 24
PY
