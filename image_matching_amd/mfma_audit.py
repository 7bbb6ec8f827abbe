"""Build-time audit of the kNN kernels' machine code: VALU writes of an MFMA's SrcC registers.

An MFMA reads its SrcC registers over its passes; the CDNA3 / CDNA4 ISA asks for (passes - 1) wait states between an
XDL MFMA and a VALU instruction that WRITES one of its SrcC registers (SrcC != vDst: an in-place accumulation is ordered
by the pipe itself). v_mfma_i32_16x16x64_i8 is a 4-pass instruction on gfx950 (16 cycles): 3 wait states, and that is
what hipcc 7.2 inserts. Nothing in the source controls where the register allocator puts the accumulator's start
values, and csrc/knn.hip's block loop re-initialises them right behind the MFMA that consumed them (the built kernels
hold 121 such writes 3 to 5 wait states behind their MFMA), so the rule is checked on what was built, independently of
the compiler: build.py disassembles the gfx950 code object of knn.o and fails the build when such a write follows its
MFMA by fewer than MIN_WAIT wait states on any straight-line or branch-taken path.

History. Round 2 blamed wrong rows 2 and 3 of a block in one experiment on a write 3-4 wait states behind the MFMA and
concluded that three wait states are not enough. Measured on the hardware in round 3 (tools/mfma_war_probe.hip,
profiles/r3_mfma_war_probe.txt: the instruction pair in inline assembly, where nothing is inserted, 0 to 8 wait states,
each of the four SrcC registers, 0 to 3 independent MFMAs queued in front, four waves per SIMD): not one wrong value
in 67 M per case, not even back to back -- the pipe interlocks or reads SrcC at issue. Three wait states are therefore
safe with room to spare, the committed kernels' 121 writes are harmless (as every exactness test says), and that
experiment's failure had another cause, which cannot be recovered (its source was not kept). The audit stays as a
cheap check of the documented rule.

The audit reads either `llvm-objdump -d` output or a compiler `.s` file. It is host-side text processing: it runs in
this container and needs no GPU. Usage: python -m image_matching_amd.mfma_audit file.{o,so,s,dis} [min_wait]"""
import os
import re
import subprocess
import sys
import tempfile

# the ISA rule for a 4-pass XDL instruction (passes - 1); the same for every register of SrcC
MIN_WAIT = 3
MIN_WAIT_FIRST = 3
LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def _regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return [int(m.group(1))] if m else []


def parse(text):
    """-> list of (kernel, address or None, mnemonic, operands list, branch target address or None)"""
    ins = []
    kern = None
    for raw in text.splitlines():
        addr = target = None
        m = re.match(r"^[0-9a-fA-F]+ <([^>]+)>:\s*$", raw)           # objdump label
        if m:
            kern = m.group(1)
            continue
        cm = re.search(r"//\s*([0-9A-Fa-f]{8,}):", raw)                # objdump: // address: encoding [<sym+off>]
        if cm:
            addr = int(cm.group(1), 16)
            tm = re.search(r"<[^>]+\+0x([0-9a-fA-F]+)>\s*$", raw)
            km = re.search(r"<([^>+]+)>\s*$", raw)
            if tm:
                target = ("off", int(tm.group(1), 16))
            elif km:
                target = ("off", 0)
        ln = raw.split("//")[0].split(";")[0].strip()
        m = re.match(r"^([A-Za-z_$.][\w$.]*):$", ln)                    # .s label
        if m:
            if not m.group(1).startswith(".L"):
                kern = m.group(1)
            ins.append((kern, None, "label", [m.group(1)], None))
            continue
        if not ln or ln.startswith("."):
            continue
        parts = ln.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if op.startswith("s_cbranch") or op == "s_branch":
            if target is None and ops:
                target = ("label", ops[0])
        else:
            target = None
        ins.append((kern, addr, op, ops, target))
    return ins


def _wait_states(op, ops):
    if op == "s_nop":
        try:
            return 1 + int(ops[0], 0)
        except (ValueError, IndexError):
            return 1
    return 1


def audit(text, min_wait=MIN_WAIT, min_wait_first=MIN_WAIT_FIRST):
    """-> list of findings (kernel, mfma line, wait states, writer line)"""
    ins = parse(text)
    # branch targets: objdump gives symbol + offset -> address of the kernel's first instruction + offset
    kern_base = {}
    by_addr = {}
    by_label = {}
    for i, (k, addr, op, ops, tgt) in enumerate(ins):
        if addr is not None:
            kern_base.setdefault(k, addr)
            by_addr[addr] = i
        if op == "label":
            by_label[ops[0]] = i
    findings = []

    def resolve(i):
        k, addr, op, ops, tgt = ins[i]
        if tgt is None:
            return None
        if tgt[0] == "off":
            return by_addr.get(kern_base.get(k, 0) + tgt[1])
        return by_label.get(tgt[1])

    def line(i):
        return "%s %s" % (ins[i][2], ", ".join(ins[i][3]))

    for i, (k, addr, op, ops, tgt) in enumerate(ins):
        if not op.startswith("v_mfma") or len(ops) < 4:
            continue
        d, c = _regs(ops[0]), _regs(ops[3])
        if not c or c == d:
            continue
        need = {r: (min_wait_first if r == c[0] else min_wait) for r in c}
        # walk every path from the MFMA until min_wait wait states have passed: (index, wait states so far)
        work = [(i + 1, 0)]
        seen = set()
        while work:
            j, w = work.pop()
            while j < len(ins) and w < min_wait:
                if (j, w) in seen:
                    break
                seen.add((j, w))
                k2, _, op2, ops2, _ = ins[j]
                if k2 != k:
                    break
                if op2 == "label":
                    j += 1
                    continue
                if op2 == "s_endpgm":
                    break
                # w wait states lie between the MFMA and this instruction
                if op2.startswith("v_") and not op2.startswith("v_cmp") and not op2.startswith("v_mfma") and ops2:
                    hit = [r for r in _regs(ops2[0]) if r in need and w < need[r]]
                    if hit:
                        findings.append((k, line(i), w, line(j)))
                if op2.startswith("s_cbranch") or op2 == "s_branch":
                    t = resolve(j)
                    if t is not None:
                        work.append((t, w + 1))
                    if op2 == "s_branch":
                        break
                w += _wait_states(op2, ops2)
                j += 1
    # one finding per (mfma, writer) pair
    out, keys = [], set()
    for f in findings:
        if (f[0], f[1], f[3]) not in keys:
            keys.add((f[0], f[1], f[3]))
            out.append(f)
    return out


def disassemble(obj_path):
    """gfx950 device code of a hipcc host object (or of libimatch.so) as llvm-objdump text."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        co = os.path.join(td, "dev.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj_path, fat])
        subprocess.check_call([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--type=o", "--unbundle",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
        return subprocess.check_output([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", co]).decode()


def audit_object(obj_path, min_wait=MIN_WAIT, min_wait_first=MIN_WAIT_FIRST):
    return audit(disassemble(obj_path), min_wait, min_wait_first)


def format_findings(findings):
    return "\n".join("%s\n   %s\n   -> +%d wait states: %s" % (k, m, w, wr) for k, m, w, wr in findings)


if __name__ == "__main__":
    path = sys.argv[1]
    mw = int(sys.argv[2]) if len(sys.argv) > 2 else MIN_WAIT
    mf = int(sys.argv[3]) if len(sys.argv) > 3 else MIN_WAIT_FIRST
    text = disassemble(path) if path.endswith((".o", ".so")) else open(path).read()
    found = audit(text, mw, mf)
    print(format_findings(found))
    print("flagged:", len(found))
    sys.exit(1 if found else 0)
