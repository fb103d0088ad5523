"""Development aid (no GPU): what the compiler left INSIDE the loops of a kernel — per loop: instructions, DPP multiply-adds,
scratch (spill) loads/stores, s_waitcnt vmcnt(0), global loads/stores, IEEE division chains.

A spill reload inside a loop that also prefetches is the thing to look for: vmcnt counts loads and stores in order, so the
wait behind the reload also waits for every prefetch in flight.

    cd <pkg>/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNMPC_SRC_HASH='"x"' -DNMPC_COL_ONLY_M=6 \
        -c nmpc_solve_col.hip -o /tmp/col.o --save-temps=obj
    python tools/asm_loops.py /tmp/nmpc_solve_col-hip-amdgcn-amd-amdhsa-gfx950.s _ZN4nmpc16solve_col_kernelILi6ELi0 [--all]
"""
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
show_all = "--all" in sys.argv
f = open(path).read().split("\n")
start = [i for i, l in enumerate(f) if l.startswith(sym)][0]
end = [i for i, l in enumerate(f) if i > start and l.startswith(".Lfunc_end")][0]
labels = {}
for i in range(start, end):
    m = re.match(r"(\.LBB\d+_\d+):", f[i].strip())
    if m:
        labels[m.group(1)] = i
loops = set()
for i in range(start, end):
    m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)|\s*s_branch\s+(\.LBB\d+_\d+)", f[i])
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            loops.add((labels[t], i))


def is_inst(s):
    s = s.strip()
    return bool(s) and not s.startswith(".") and not s.startswith(";") and not s.endswith(":")


for a, b in sorted(loops):
    inner = not any(a2 >= a and b2 <= b and (a2, b2) != (a, b) for a2, b2 in loops)
    if not (inner or show_all):
        continue
    body = [f[i].strip() for i in range(a, b + 1) if is_inst(f[i])]
    cnt = lambda p: sum(s.startswith(p) for s in body)
    vm0 = sum(s.startswith("s_waitcnt") and "vmcnt(0)" in s for s in body)
    print("%sloop %5d-%5d: %4d instr, dpp %3d, global ld %2d st %2d, scratch ld %2d st %2d, vmcnt(0) %2d, divisions %d"
          % ("inner " if inner else "      ", a - start, b - start, len(body), sum("_dpp" in s for s in body), cnt("global_load"), cnt("global_store"),
             cnt("scratch_load"), cnt("scratch_store"), vm0, cnt("v_div_fmas")))
