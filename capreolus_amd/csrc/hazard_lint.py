"""Build-time check of the gfx950 code for the one hazard the assembler cannot see for us.

The ring GEMMs issue their MFMAs as inline assembly (bert_gemm_ring16.h: Mfma16, "+a" accumulators) so that the register allocator
leaves the 256 accumulator registers where they are.  The price: LLVM's hazard recogniser does not look inside an inline-assembly
block, so it inserts no wait states between such an MFMA and a LATER, compiler-generated instruction that touches the MFMA's result -
an accumulator it decided to copy or spill (v_accvgpr_read / v_accvgpr_mov) right behind the instruction that is still writing it.
The copy then holds the value from before the MFMA: wrong scores, no fault, and only in the build whose register pressure tipped over
(round 5: a 16-register prefetch made the -DCAPAMD_PROFILING build of the QKV kernel spill four tiles inside its K loop; the product
build of the same source was correct).

This script disassembles the gfx950 code object embedded in each object file and walks every function in program order: after each
MFMA it counts wait states (one per instruction, N + 1 for `s_nop N`) and reports any non-MFMA instruction that names a register of
the MFMA's destination before `passes + 2` of them have gone by (the matrix-write -> VALU/VMEM/LDS-access rule; calibrated on what the
compiler itself leaves behind the MFMAs it can see: 10 behind an 8-pass 16x16x4 f32, 12 behind an 8-pass 32x32x16).  The walk follows
branch targets (forward and backward: lint_listing).  Code the compiler scheduled itself passes by construction; a finding means an
inline-assembly MFMA's result is being touched too early (any operand position: reads and overwrites alike).  build.py runs it over
EVERY object it links - rebuilt in this run or not - and fails the build on a finding, or when it cannot run (no llvm-objdump).
Not covered: an MFMA's SOURCE registers overwritten while it still reads them (the ring kernels' A / B fragments are only ever
rewritten by ds_read results, a round trip later; srcC is the destination itself or the literal 0).
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

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


kTakenBranchStates = 2      # wait states a taken branch is worth (lint_listing: over_edge)


def _passes(mnemonic):
    """4-cycle passes of an MFMA on gfx950 (the 16-bit 16x16x32 / 32x32x16 forms run at twice the gfx942 rate: 4 and 8)"""
    if "16x16x32" in mnemonic:
        return 4
    if "32x32x16" in mnemonic or "16x16" in mnemonic:
        return 8
    if "32x32" in mnemonic:
        return 16
    return 2


_ADDR = re.compile(r"//\s*([0-9A-Fa-f]{8,}):")
_TARGET = re.compile(r"<([^<>]+)\+0x([0-9a-fA-F]+)>\s*$|<([^<>+]+)>\s*$")
_HEAD = re.compile(r"^([0-9a-f]+) <(.+)>:$")
_LABEL_TARGET = re.compile(r"^(s_c?branch\w*)\s+(\.?[A-Za-z_][\w.$]*)\s*$")


def _parse(lines):
    """[(function, address or None, instruction text, branch target address / label or None)] of a disassembly (llvm-objdump -d: every
    line carries its address in the trailing comment, a branch its target as <function+0xoffset>) or of a compiler .s listing (labels)"""
    out, func, starts, labels = [], "?", {}, {}
    for raw in lines:
        head = _HEAD.match(raw.strip())
        if head:
            func = head.group(2)
            starts[func] = int(head.group(1), 16)
            continue
        code, _, comment = raw.partition("//")
        line = code.rstrip()
        if line.endswith(":") and not line.startswith(("\t", " ")):
            name = line[:-1]
            labels[name] = len(out)
            if not name.startswith(".L"):
                func = name
            continue
        ins = line.strip()
        if not ins or ins.startswith((";", ".", "//")):
            continue
        m = _ADDR.search("//" + comment) if comment else None
        addr = int(m.group(1), 16) if m else None
        target = None
        mnem = ins.split()[0]
        if mnem.startswith(("s_cbranch", "s_branch")):
            t = _TARGET.search(comment) if comment else None
            if t and (t.group(1) or t.group(3)):
                name = t.group(1) or t.group(3)
                off = int(t.group(2), 16) if t.group(2) else 0
                if name in starts:
                    target = starts[name] + off
            else:
                lt = _LABEL_TARGET.match(ins)
                if lt:
                    target = lt.group(2)
        out.append((func, addr, ins, target))
    index = {a: i for i, (_, a, _, _) in enumerate(out) if a is not None}
    index.update(labels)
    return out, index


def _step(func, ins, live, findings):
    """One instruction against the MFMA results still owed wait states; returns the new live list."""
    mnem = ins.split()[0]
    if mnem.startswith("v_mfma") or mnem.startswith("v_smfmac"):
        for e in live:
            e[1] -= 1
            e[2] += 1
        live = [e for e in live if e[1] > 0]
        dest = ins[len(mnem):].split(",")[0]
        live.append([_regs(dest), _passes(mnem) + 2, 0, ins])
        return live
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
    return [e for e in live if e[1] > 0]


def lint_listing(lines):
    """[(function, mfma line, offending line, wait states seen, needed)] over a disassembly / assembly listing.

    The walk follows the program's edges, not only its text: the state behind an MFMA - which result registers are still owed how many
    wait states - is carried (a) to the next instruction, except behind s_branch / s_endpgm / s_setpc, and (b) to the TARGET of every
    branch, conditional or not, forward or backward (a loop's back edge carries the tail's MFMAs to the loop header), where a bounded
    walk continues until nothing is owed any more (an MFMA's debt is at most 18 wait states).  Up to round 5 the walk was linear: a
    read at a branch target was never looked at."""
    prog, index = _parse(lines)
    findings, seen = [], set()

    def over_edge(live):
        # a TAKEN branch is not one wait state: the wave's instruction buffer is refilled from the target.  Calibrated like the rest of
        # this file on what hipcc leaves behind MFMAs it can see: the tightest such path in the libraries (cedr_pool_cm_kernel: an 8-pass
        # 32x32x16, s_cbranch_vccnz, `s_nop 6` at the target, v_accvgpr_read of the result) has 8 counted states where 10 are asked for
        # in line - the edge is credited with the difference, kTakenBranchStates = 2
        out = []
        for e in live:
            if e[1] - kTakenBranchStates > 0:
                out.append([set(e[0]), e[1] - kTakenBranchStates, e[2] + kTakenBranchStates, e[3]])
        return out

    def walk(start, live, depth):
        i, budget = start, 96
        while live and 0 <= i < len(prog) and budget > 0:
            func, _, ins, target = prog[i]
            mnem = ins.split()[0]
            live = _step(func, ins, live, findings)
            budget -= 1
            if target is not None and live and depth < 4 and target in index:
                key = (index[target], tuple(sorted((tuple(sorted(e[0])), e[1]) for e in live)))
                if key not in seen:
                    seen.add(key)
                    walk(index[target], over_edge(live), depth + 1)
            if mnem in ("s_branch", "s_endpgm", "s_setpc_b64"):
                return
            i += 1

    live, func = [], None
    for i, (f, _, ins, target) in enumerate(prog):
        if f != func:
            func, live = f, []
        mnem = ins.split()[0]
        live = _step(f, ins, live, findings)
        if target is not None and live and target in index:
            key = (index[target], tuple(sorted((tuple(sorted(e[0])), e[1]) for e in live)))
            if key not in seen:
                seen.add(key)
                walk(index[target], over_edge(live), 1)
        if mnem in ("s_branch", "s_endpgm", "s_setpc_b64"):
            live = []
    # (the same finding can be reached along several edges)
    uniq, out = set(), []
    for f in findings:
        if f[:3] not in uniq:
            uniq.add(f[:3])
            out.append(f)
    return out


def objdump():
    """llvm-objdump of the ROCm installation hipcc comes from (ROCM_PATH, the directory of hipcc, /opt/rocm); None when there is none"""
    import shutil

    roots = [os.environ.get("ROCM_PATH"), os.environ.get("HIP_PATH")]
    hipcc = shutil.which("hipcc")
    if hipcc:
        roots.append(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))))
    roots.append("/opt/rocm")
    for r in roots:
        if r:
            for cand in (os.path.join(r, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(r, "llvm", "bin", "llvm-objdump")):
                if os.path.exists(cand):
                    return cand
    return shutil.which("llvm-objdump")


def lint_object(path):
    tool = objdump()
    if tool is None:
        raise FileNotFoundError("llvm-objdump not found (ROCM_PATH / hipcc's installation / /opt/rocm): the MFMA hazard lint cannot run")
    findings = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            dis = subprocess.run([tool, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
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
