#!/usr/bin/env python3
"""Static hazard check of the VGPR-form (inline asm) MFMAs in the gfx950 code objects of libmi355fa.so.

    python tools/mfma_lint.py [path/to/lib.so] [-v]      -> violations per kernel, exit status 1 if any

hipcc pads MFMA hazards for its own MFMA builtins only.  The one-wave-per-SIMD kernels (fa_bwd_dkv_v3.hip,
fa_fwd_v4.hip, fa_bwd_dq_v4.hip) issue their score chains as inline-asm MFMAs with an architectural-VGPR accumulator
(fa_common.h mfma_v_* / MfmaPin), which the hazard recognizer does not see -- and the register allocator is free to put a
copy, or a reused register, right next to one.  Both failures have been met on the GPU (a row constant's registers reused
for exp results under the MFMA that reads them: run-to-run 1-ulp differences; an operand copied into place by
v_accvgpr_write directly in front of the MFMA: a wrong row block), both silent.  This tool disassembles every kernel
(llvm-objdump) and checks every MFMA whose destination is a VGPR range (hipcc's own accumulate in AGPRs in these kernels;
where they do not, hipcc padded them and the rules hold anyway) on a simple issue-time model, in wait states (one
instruction = 1, a transcendental or an MFMA = 2, s_nop N = N + 1, an MFMA issues no earlier than 8 after the previous one: v_mfma_f32_32x32x16 is 8 passes):

  R1  no VALU instruction writes a register of the MFMA's A, B or C operand less than 2 wait states before it
  R2  nothing writes a register of its C operand (where C is not D) before the NEXT MFMA has issued
  R3  nothing but an MFMA reads or writes its D registers earlier than 12 wait states after it
  R5  no spill-lane restore (v_readlane / v_readfirstlane) of a descriptor or scalar offset less than 5 wait states before an
      inline-asm buffer load that reads it (LDS-DMA, prefetch into accumulator registers): a stale descriptor is a memory fault
  R4  the pinned accumulator file a[0:193] of fa_common.h is touched only by the instructions that own it: accumulators by
      MFMAs, zeroing v_accvgpr_write_b32 and the epilogue's v_accvgpr_read_b32; resident operands by sets of v_accvgpr_write_b32,
      MFMAs and (the scaled-Q workspace store) sets of 32 v_accvgpr_read_b32; prefetch registers by buffer_load_dword and
      v_accvgpr_read_b32 -- hipcc never allocates any of them

The model is linear (it follows the instruction stream, not branches): a hazard across a taken branch is not seen.
Not part of the product; tests/test_codeobj.py runs it.
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codeobj  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out



def regs(text):
    """set of ('v'|'a', index) named in an operand string"""
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


class Inst:
    __slots__ = ("op", "ops", "writes", "reads", "is_mfma", "is_valu", "ws")

    def __init__(self, op, operands):
        self.op, self.ops = op, operands
        self.is_mfma = op.startswith("v_mfma")
        self.is_valu = op.startswith("v_") and not self.is_mfma
        # issue slots: a transcendental and an MFMA hold the vector issue port for 8 cycles, everything else for <= 4
        self.ws = 2 if (self.is_mfma or op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos"))) else 1
        if op == "s_nop":
            self.ws = int(operands[0], 0) + 1 if operands else 1
        nw = 0   # leading operands that are written
        if op.startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
            nw = 0   # write SGPRs / VCC only
        elif op.startswith(("v_permlane32_swap", "v_permlane16_swap", "v_swap")):
            nw = 2
        elif op.startswith("v_"):
            nw = 1
        elif op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"):
            nw = 1
        elif op.startswith(("buffer_load", "global_load", "flat_load", "scratch_load")) and "lds" not in operands:
            nw = 1
        self.writes, self.reads = set(), set()
        for i, o in enumerate(operands):
            (self.writes if i < nw else self.reads).update(regs(o))
        if self.is_mfma and len(operands) >= 4 and operands[0] == operands[3]:
            self.reads.update(regs(operands[0]))   # accumulate in place


def disassemble(elf):
    """{kernel symbol: [Inst]} of one code object"""
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        text = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith(("\t", " ")):
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        op, _, rest = body.partition(" ")
        operands = [o.strip() for o in rest.split(",")] if rest.strip() else []
        cur.append(Inst(op, operands))
    return out


def lint_kernel(insts):
    bad = []
    # a kernel on the pinned accumulator file issues ALL its MFMAs as inline asm (fa_common.h)
    all_asm = any(i.is_mfma and len(i.ops) >= 4 and i.ops[0].startswith("v") and regs(i.ops[2]) & {("a", k) for k in range(0, 194)} for i in insts)
    for k, m in enumerate(insts):
        if not m.is_mfma or len(m.ops) < 4:
            continue
        d, a, b, c = (regs(o) for o in m.ops[:4])
        if not d:
            continue
        if any(f != "v" for f, _ in d):
            if not all_asm:
                continue   # AGPR accumulator: hipcc's own MFMA, hipcc's padding
            c = set()      # a pinned accumulator of an all-asm kernel: only R1 on A / B applies (its chain is MFMA-only)
            d = set()
        # R1: VALU writes of A / B / C in front
        ws, j = 0, k - 1
        while j >= 0 and ws < 2:
            p = insts[j]
            if p.is_valu and p.writes & (a | b | c):
                bad.append("R1 %s %s  <- %s %s (%d wait states before)" % (m.op, ", ".join(m.ops), p.op, ", ".join(p.ops), ws))
            ws += p.ws
            j -= 1
        # R2 / R3: after it
        t, last_mfma, seen_next = 0, 0, False
        cd = c - d
        for n in insts[k + 1:]:
            t += n.ws
            if n.is_mfma:
                t = max(t, last_mfma + 8)
                last_mfma = t
            if not seen_next and not n.is_mfma and n.writes & cd:
                bad.append("R2 %s %s  -> %s %s writes C before the next MFMA" % (m.op, ", ".join(m.ops), n.op, ", ".join(n.ops)))
            if t < 12 and not n.is_mfma and (n.writes | n.reads) & d:
                bad.append("R3 %s %s  -> %s %s touches D after %d wait states" % (m.op, ", ".join(m.ops), n.op, ", ".join(n.ops), t))
            if n.is_mfma:
                seen_next = True
            if t >= 12 and seen_next:
                break
    # R5: an inline-asm buffer load (LDS-DMA, or an accumulator-register destination: hipcc emits neither) whose descriptor or
    # scalar offset was written by a VALU instruction (v_readlane / v_readfirstlane: a spill-lane restore) < 5 wait states before
    for k, m in enumerate(insts):
        if not m.op.startswith("buffer_load") or not ("lds" in m.ops or (m.ops and m.ops[0].startswith("a"))):
            continue
        need = set()
        for o in m.ops[1:]:
            need |= sregs(o)
        ws, j = 0, k - 1
        while j >= 0 and ws < 5:
            q = insts[j]
            if q.op.startswith(("v_readlane", "v_readfirstlane")) and q.ops and sregs(q.ops[0]) & need:
                bad.append("R5 %s %s  <- %s %s (%d wait states before)" % (m.op, ", ".join(m.ops), q.op, ", ".join(q.ops), ws))
            ws += q.ws
            j -= 1
    # R4: the pinned accumulator file a[0:193] (fa_common.h).  A kernel uses it iff one of its VGPR-form MFMAs takes its B
    # operand from an accumulator register; the dQ kernel keeps accumulators a[0:63], resident operands a[64:127] and prefetch
    # registers a[128:129], the dK/dV kernel a[0:127], a[128:191] and a[192:193].
    pinned = {("a", i) for i in range(0, 194)}
    uses = [i for i in insts if i.is_mfma and len(i.ops) >= 4 and i.ops[0].startswith("v") and regs(i.ops[2]) & pinned]
    if uses:
        lo = min(n for i in uses for f, n in regs(i.ops[2]) if f == "a")
        n_acc, n_res = (64, 64) if lo < 128 else (128, 64)
        acc = {("a", i) for i in range(0, n_acc)}
        res = {("a", i) for i in range(n_acc, n_acc + n_res)}
        pf = {("a", i) for i in range(n_acc + n_res, n_acc + n_res + 2)}
        rest = pinned - acc - res - pf

        def ops_of(sel):
            return sorted({i.op for i in sel})
        wr = [i for i in insts if i.writes & acc and not i.is_mfma]
        if any(i.op != "v_accvgpr_write_b32" for i in wr) or len(wr) % n_acc:
            bad.append("R4 accumulators written by %d non-MFMA instructions (%s), expected sets of %d zeroing v_accvgpr_write_b32" % (len(wr), ops_of(wr), n_acc))
        rd = [i for i in insts if i.reads & acc and not i.is_mfma]
        if any(i.op != "v_accvgpr_read_b32" for i in rd) or len(rd) % n_acc:
            bad.append("R4 accumulators read by %d non-MFMA instructions (%s), expected sets of %d v_accvgpr_read_b32" % (len(rd), ops_of(rd), n_acc))
        wr = [i for i in insts if i.writes & res]
        if not wr or len(wr) % n_res or any(i.op != "v_accvgpr_write_b32" for i in wr):
            bad.append("R4 resident operands written by %d instructions (%s), expected sets of %d v_accvgpr_write_b32" % (len(wr), ops_of(wr), n_res))
        rd = [i for i in insts if i.reads & res and not i.is_mfma]
        if len(rd) % 32 or any(i.op != "v_accvgpr_read_b32" for i in rd):
            bad.append("R4 resident operands read by %d non-MFMA instructions (%s)" % (len(rd), ops_of(rd)))
        wr = [i for i in insts if i.writes & pf]
        if any(i.op != "buffer_load_dword" for i in wr):
            bad.append("R4 prefetch registers written by %s, expected buffer_load_dword only" % ops_of(wr))
        rd = [i for i in insts if i.reads & pf]
        if any(i.op != "v_accvgpr_read_b32" for i in rd):
            bad.append("R4 prefetch registers read by %s, expected v_accvgpr_read_b32 only" % ops_of(rd))
        other = [i for i in insts if (i.reads | i.writes) & rest]
        if other:
            bad.append("R4 %d instructions (%s) touch pinned registers this kernel does not use" % (len(other), ops_of(other)))
    return bad


def lint(lib_path=codeobj.DEFAULT_LIB):
    """{short kernel name: [violations]} over every kernel of the library that has a VGPR-form MFMA or pinned registers"""
    out = {}
    for co in codeobj.code_objects(lib_path):
        for name, insts in disassemble(co).items():
            if not name.startswith("_ZN2fa"):
                continue
            n_asm = sum(1 for i in insts if i.is_mfma and i.ops and i.ops[0].startswith("v"))
            n_acc = sum(1 for i in insts if i.is_mfma and i.ops and i.ops[0].startswith("a"))
            # hipcc picks ONE register form for all MFMA builtins of a kernel: VGPR-form MFMAs beside AGPR-form ones are the
            # inline-asm ones; a kernel with VGPR-form MFMAs only is hipcc's own choice and hipcc's own padding
            if n_asm == 0 or n_acc == 0:
                continue
            out[codeobj.demangle_short(name)] = (n_asm, lint_kernel(insts))
    return out


if __name__ == "__main__":
    args = [x for x in sys.argv[1:] if not x.startswith("-")]
    res = lint(args[0] if args else codeobj.DEFAULT_LIB)
    total = 0
    for name in sorted(res):
        n_asm, bad = res[name]
        total += len(bad)
        print("%-44s %4d VGPR-form MFMAs  %s" % (name, n_asm, "ok" if not bad else "%d violations" % len(bad)))
        if "-v" in sys.argv or bad:
            for b in bad[:12]:
                print("    " + b)
    sys.exit(1 if total else 0)
