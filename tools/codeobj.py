#!/usr/bin/env python3
"""Kernel resource metadata of the gfx950 code objects embedded in libmi355fa.so.

    python tools/codeobj.py [path/to/lib.so]      -> one line per kernel: VGPRs, AGPRs, SGPRs, spills, scratch, LDS

The shared library carries one clang offload bundle per translation unit in its .hip_fatbin section.  This reads the
section straight out of the ELF (no GPU, no HIP runtime), unpacks every gfx950 entry and parses the AMDGPU metadata note
(msgpack) of each code object.  tests/test_codeobj.py uses `kernels()` to fail the CPU suite on register spills.
Not part of the product; nothing in the package imports it.
"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "flashattention-from-scratch-with-triton_amd", "libmi355fa.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_sections(blob):
    """(name, offset, size) of every section of a 64-bit little-endian ELF image."""
    assert blob[:4] == b"\x7fELF" and blob[4] == 2 and blob[5] == 1, "not a 64-bit LE ELF"
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", blob, shoff + i * shentsize)
        secs.append((name, typ, off, size))
    stro, strs = secs[shstrndx][2], secs[shstrndx][3]
    strtab = blob[stro:stro + strs]

    def nm(o):
        return strtab[o:strtab.index(b"\0", o)].decode()
    return [(nm(n), t, o, s) for n, t, o, s in secs]


def code_objects(lib_path, arch="gfx950"):
    """ELF images of every `arch` device code object bundled into the host shared library."""
    blob = open(lib_path, "rb").read()
    out = []
    for name, _typ, off, size in _elf_sections(blob):
        if name != ".hip_fatbin":
            continue
        fat = blob[off:off + size]
        pos = fat.find(MAGIC)
        while pos >= 0:
            n, = struct.unpack_from("<Q", fat, pos + len(MAGIC))
            p = pos + len(MAGIC) + 8
            for _ in range(n):
                eoff, esize, tsize = struct.unpack_from("<QQQ", fat, p)
                triple = fat[p + 24:p + 24 + tsize].decode()
                p += 24 + tsize
                if arch in triple and esize:
                    out.append(fat[pos + eoff:pos + eoff + esize])
            pos = fat.find(MAGIC, pos + 1)
    return out


def _msgpack(b, i=0):
    """Minimal msgpack decoder (maps, arrays, strings, ints, bools, nil) -- enough for AMDGPU metadata."""
    t = b[i]
    if t <= 0x7F:
        return t, i + 1
    if 0x80 <= t <= 0x8F:
        return _map(b, i + 1, t & 15)
    if 0x90 <= t <= 0x9F:
        return _arr(b, i + 1, t & 15)
    if 0xA0 <= t <= 0xBF:
        n = t & 31
        return b[i + 1:i + 1 + n].decode(), i + 1 + n
    if t == 0xC0:
        return None, i + 1
    if t in (0xC2, 0xC3):
        return t == 0xC3, i + 1
    if t in (0xC4, 0xD9):
        n = b[i + 1]
        return b[i + 2:i + 2 + n].decode(errors="replace"), i + 2 + n
    if t in (0xC5, 0xDA):
        n, = struct.unpack_from(">H", b, i + 1)
        return b[i + 3:i + 3 + n].decode(errors="replace"), i + 3 + n
    if t in (0xC6, 0xDB):
        n, = struct.unpack_from(">I", b, i + 1)
        return b[i + 5:i + 5 + n].decode(errors="replace"), i + 5 + n
    if t == 0xCC:
        return b[i + 1], i + 2
    if t == 0xCD:
        return struct.unpack_from(">H", b, i + 1)[0], i + 3
    if t == 0xCE:
        return struct.unpack_from(">I", b, i + 1)[0], i + 5
    if t == 0xCF:
        return struct.unpack_from(">Q", b, i + 1)[0], i + 9
    if t == 0xD0:
        return struct.unpack_from(">b", b, i + 1)[0], i + 2
    if t == 0xD1:
        return struct.unpack_from(">h", b, i + 1)[0], i + 3
    if t == 0xD2:
        return struct.unpack_from(">i", b, i + 1)[0], i + 5
    if t == 0xD3:
        return struct.unpack_from(">q", b, i + 1)[0], i + 9
    if t == 0xDC:
        return _arr(b, i + 3, struct.unpack_from(">H", b, i + 1)[0])
    if t == 0xDD:
        return _arr(b, i + 5, struct.unpack_from(">I", b, i + 1)[0])
    if t == 0xDE:
        return _map(b, i + 3, struct.unpack_from(">H", b, i + 1)[0])
    if t == 0xDF:
        return _map(b, i + 5, struct.unpack_from(">I", b, i + 1)[0])
    if t >= 0xE0:
        return t - 256, i + 1
    raise ValueError("msgpack type 0x%02x" % t)


def _map(b, i, n):
    d = {}
    for _ in range(n):
        k, i = _msgpack(b, i)
        v, i = _msgpack(b, i)
        d[k] = v
    return d, i


def _arr(b, i, n):
    a = []
    for _ in range(n):
        v, i = _msgpack(b, i)
        a.append(v)
    return a, i


def metadata(elf):
    """The amdhsa metadata map (NT_AMDGPU_METADATA, type 32) of one device code object."""
    for name, typ, off, size in _elf_sections(elf):
        if typ != 7:  # SHT_NOTE
            continue
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            nname = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32 and nname.startswith(b"AMDGPU"):
                return _msgpack(desc)[0]
    return {}


def kernels(lib_path=DEFAULT_LIB):
    """[{name, vgpr, agpr, sgpr, spill, scratch, lds, wg}] for every gfx950 kernel in the library."""
    out = []
    for co in code_objects(lib_path):
        for k in metadata(co).get("amdhsa.kernels", []):
            out.append({"name": k[".name"], "vgpr": k.get(".vgpr_count", 0), "agpr": k.get(".agpr_count", 0),
                        "sgpr": k.get(".sgpr_count", 0), "spill": k.get(".vgpr_spill_count", 0),
                        "sgpr_spill": k.get(".sgpr_spill_count", 0),
                        "scratch": k.get(".private_segment_fixed_size", 0),
                        "lds": k.get(".group_segment_fixed_size", 0), "wg": k.get(".max_flat_workgroup_size", 0)})
    return out


def demangle_short(name):
    """fa_bwd_dq_kernel<64,BF16,true,3> style label from the mangled template instance."""
    import re
    m = re.match(r"_ZN2fa\d+(\w+?_kernel)I(.*)EvNS_\d+\w+E$", name)
    if not m:
        return name
    args = []
    for kind, val, typ in re.findall(r"L([bij])(\d+)E|NS_\d+([A-Z0-9]+)E", m.group(2)):
        if typ:
            args.append(typ)
        else:
            args.append(("true" if val == "1" else "false") if kind == "b" else val)
    return "%s<%s>" % (m.group(1), ",".join(args))


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB
    ks = sorted(kernels(path), key=lambda k: k["name"])
    print("%-52s %5s %5s %5s %6s %8s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "spill", "scratch", "lds"))
    for k in ks:
        print("%-52s %5d %5d %5d %6d %8d %7d" % (demangle_short(k["name"]), k["vgpr"], k["agpr"], k["sgpr"],
                                                 k["spill"], k["scratch"], k["lds"]))
