"""Development aid / build check (no GPU): the v_fmac_f64_dpp inline asm of the column kernel reads lanes of its first source through
the DPP path, which needs 2 wait states after a VALU write of that register (gfx9 "VALU writes VGPR -> DPP reads that VGPR").  The
compiler's hazard recognizer does not look inside inline asm; the kernel places `s_nop` markers itself and relies on the register
allocator inserting no copy between marker and consumer.  This script checks the property on the compiled kernel: for every *_dpp
instruction, walk back over 2 wait states (s_nop N counts N + 1, any other instruction 1) and report a VALU / permlane-swap write of
the DPP source register inside that window.

    python tools/asm_hazards.py <kernel.s> [symbol prefix]        -> prints "<n dpp> <n hazards>", exit code 1 on a hazard
"""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(path, prefix=""):
    lines = open(path).read().split("\n")
    n_dpp = 0
    hazards = []
    hist = []      # (opcode, set of VGPRs written by a VALU op, wait states the instruction provides)
    infn = False
    for ln in lines:
        t = ln.split(";")[0].strip()
        if not t or t.startswith("."):
            if t.startswith(".Lfunc_end"):
                infn = False
            continue
        if t.endswith(":"):
            if not t.startswith(".L"):
                infn = t.startswith(prefix)
                hist = []
            else:
                hist.append(("label", set(), 0))
            continue
        if not infn:
            continue
        parts = t.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if "_dpp" in op:
            n_dpp += 1
            src = regs(ops[1].split()[0]) if len(ops) > 1 else set()
            ws = 0
            for pop, pw, pws in reversed(hist):
                if ws >= 2:
                    break
                if pop == "label":      # a join: the other predecessor is checked when its own fall-through is walked; be strict here
                    continue
                if pw & src:
                    hazards.append((op, sorted(src), pop, ws))
                    break
                ws += pws
        written = set()
        if op.startswith("v_") and ops:
            written = regs(ops[0].split()[0])
            if "permlane" in op and "swap" in op and len(ops) > 1:
                written |= regs(ops[1].split()[0])
        wsn = 1
        if op == "s_nop":
            wsn = int(ops[0], 0) + 1
        hist.append((op, written, wsn))
        if len(hist) > 16:
            hist.pop(0)
    return n_dpp, hazards


if __name__ == "__main__":
    n, hz = check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    print(n, len(hz))
    for h in hz[:20]:
        print("  hazard:", h)
    sys.exit(1 if hz else 0)
