"""Build-time check of the gfx950 code for the one hazard the assembler cannot see for us.

The ring GEMMs issue their MFMAs as inline assembly (bert_gemm_ring16.cuh: Mfma16, "+a" accumulators) so that the register allocator
leaves the 256 accumulator registers where they are.  The price: LLVM's hazard recogniser does not look inside an inline-assembly
block, so it inserts no wait states between such an MFMA and a LATER, compiler-generated instruction that touches the MFMA's result -
an accumulator it decided to copy or spill (v_accvgpr_read / v_accvgpr_mov) right behind the instruction that is still writing it.
The copy then holds the value from before the MFMA: wrong scores, no fault, and only in the build whose register pressure tipped over
(round 5: a 16-register prefetch made the -DCAPAMD_PROFILING build of the QKV kernel spill four tiles inside its K loop; the product
build of the same source was correct).

This script disassembles the gfx950 code object embedded in each object file and walks every function in program order: after each
MFMA it counts wait states (one per instruction, N + 1 for `s_nop N`) and reports any non-MFMA instruction that names a register of
the MFMA's destination before `passes + 2` of them have gone by (the matrix-write -> VALU/VMEM/LDS-access rule; calibrated on what the
compiler itself leaves behind the MFMAs it can see: 10 behind an 8-pass 16x16x4 f32, 12 behind an 8-pass 32x32x16).  The walk is
linear and starts afresh behind an unconditional branch.  Code the compiler scheduled itself passes by construction; a finding means
an inline-assembly MFMA's result is being read too early.  build.py runs it over every object it links and fails the build on a finding.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_REG = re.compile(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b")
_FUNC = re.compile(r"^[0-9a-f]+ <(.+)>:$")


def code_objects(path, arch="gfx950"):
    """The device code objects for `arch` bundled into a host object / shared library (clang offload bundle, uncompressed)"""
    blob = open(path, "rb").read()
    out, at = [], 0
    while True:
        i = blob.find(_MAGIC, at)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tl].decode()
            p += tl
            if arch in triple and size:
                out.append(blob[i + off:i + off + size])
        at = i + len(_MAGIC)


def _regs(text):
    s = set()
    for m in _REG.finditer(text):
        if m.group(1):
            s.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            s.add((m.group(4), int(m.group(5))))
    return s


def _passes(mnemonic):
    """4-cycle passes of an MFMA on gfx950 (the 16-bit 16x16x32 / 32x32x16 forms run at twice the gfx942 rate: 4 and 8)"""
    if "16x16x32" in mnemonic:
        return 4
    if "32x32x16" in mnemonic or "16x16" in mnemonic:
        return 8
    if "32x32" in mnemonic:
        return 16
    return 2


def lint_listing(lines):
    """[(function, mfma line, offending line, wait states seen, needed)] over a disassembly / assembly listing"""
    findings, func, live = [], "?", []          # live: [dest registers, wait states still owed, states seen, text]
    for raw in lines:
        line = raw.split("//")[0].rstrip()
        m = _FUNC.match(line.strip())
        if m:
            func, live = m.group(1), []
            continue
        if line.endswith(":") and not line.startswith(("\t", " ")):
            func, live = line[:-1], []
            continue
        ins = line.strip()
        if not ins or ins.startswith((";", ".", "//")):
            continue
        mnem = ins.split()[0]
        if mnem.startswith("v_mfma") or mnem.startswith("v_smfmac"):
            for e in live:
                e[1] -= 1
                e[2] += 1
            live = [e for e in live if e[1] > 0]
            dest = ins[len(mnem):].split(",")[0]
            live.append([_regs(dest), _passes(mnem) + 2, 0, ins])
            continue
        if mnem in ("s_branch", "s_endpgm", "s_setpc_b64"):
            live = []
            continue
        used = _regs(ins[len(mnem):]) if live else ()
        for e in live:
            if used and not e[0].isdisjoint(used):
                findings.append((func, e[3], ins, e[2], e[2] + e[1]))
        w = 1
        if mnem == "s_nop":
            w = int(ins.split()[1], 0) + 1
        for e in live:
            e[1] -= w
            e[2] += w
        live = [e for e in live if e[1] > 0]
    return findings


def lint_object(path):
    findings = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        findings += lint_listing(dis.splitlines())
    return findings


def main(argv):
    bad = 0
    for path in argv:
        if path.endswith(".s"):
            found = lint_listing(open(path).read().splitlines())
        else:
            found = lint_object(path)
        for func, mfma, ins, seen, need in found:
            print(f"{os.path.basename(path)}: {func}\n    {mfma}\n    -> {ins}   ({seen} wait states, {need} needed)")
        bad += len(found)
    print(f"hazard_lint: {bad} finding(s) in {len(argv)} file(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
