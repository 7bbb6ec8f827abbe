"""Development: looks for the write-after-read pattern that broke an experiment on this kernel -- a vector instruction
writing registers that an MFMA issued a few instructions earlier reads as SrcC (SrcC != vDst): hipcc 7.2 pads it with 3
wait states for v_mfma_i32_16x16x64_i8, and rows 2 / 3 of the block came out wrong. Lists every such write within N
instructions after the MFMA, per kernel (straight-line scan, labels ignored: an over-approximation).
Usage: python tools/dbg/mfma_war_audit.py file.s [N=10]"""
import re, sys
path = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = lambda t: (lambda m: range(int(m.group(1)), int(m.group(2)) + 1) if m else None)(re.match(r"v\[(\d+):(\d+)\]", t))
def regs(tok):
    r = rng(tok)
    if r is not None:
        return set(r)
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
kern = None
ins = []
for ln in open(path):
    ln = ln.split(";")[0].strip()
    if not ln or ln.startswith("."):
        m = re.match(r"^(_Z\w+):", ln) if ln else None
        continue
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        kern = m.group(1)
        continue
    if ln.endswith(":"):
        continue
    ins.append((kern, ln))
bad = 0
for i, (k, ln) in enumerate(ins):
    if not ln.startswith("v_mfma"):
        continue
    ops = [o.strip() for o in ln.split(None, 1)[1].split(",")]
    d, c = regs(ops[0]), regs(ops[3])
    if not c or c == d:
        continue
    slots = 0
    for k2, l2 in ins[i + 1:i + 1 + 3 * N]:
        if k2 != k:
            break
        op = l2.split()[0]
        slots += 1 + (int(l2.split()[1]) if op == "s_nop" else 0)
        if slots > N:
            break
        if op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_mfma"):
            w = regs(l2.split(None, 1)[1].split(",")[0].strip())
            if w & c:
                bad += 1
                print("%s\n   %s\n   -> +%d wait states: %s" % (k[:40], ln, slots, l2))
print("flagged:", bad)
