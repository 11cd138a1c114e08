#!/usr/bin/env python3
"""Disassembly of one kernel of the library (llvm-objdump) to a file.  usage: isa_dump.py <mangled-name substring> out.s [lib.so]"""
import re, subprocess, sys, tempfile, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codeobj
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
pat, out = sys.argv[1], sys.argv[2]
lib = sys.argv[3] if len(sys.argv) > 3 else codeobj.DEFAULT_LIB
for co in codeobj.code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        text = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
    for part in re.split(r"\n(?=[0-9a-f]+ <)", text):
        head = part.split("\n", 1)[0]
        if pat in head:
            open(out, "w").write(part)
            print(head, len(part.splitlines()), "lines ->", out)
            sys.exit(0)
sys.exit("no kernel matches " + pat)
