"""Instruction mix per LOOP of one kernel of the gfx950 assembly (hipcc -S): basic blocks are grouped by the loop header LLVM
names in its block comments ("in Loop: Header=BBx_y Depth=d"), inner loops (spin waits) listed separately.
python benchmarks/isa_loops.py <file.s> <mangled-kernel-name> [min_instrs]"""
import collections
import re
import sys

src = open(sys.argv[1]).read()
name = sys.argv[2]
thresh = int(sys.argv[3]) if len(sys.argv) > 3 else 30
start = src.index("\n" + name + ":") + 1
end = src.index(".Lfunc_end", start)
cur = ("-", 0)
counts = collections.defaultdict(collections.Counter)
for ln in src[start:end].split("\n")[1:]:
    t = ln.strip()
    m = re.match(r"^(\.LBB\d+_\d+):\s*;?\s*(.*)$", t)
    if m:
        c = m.group(2)
        h = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", c)
        if "Loop Header" in c:
            d = re.search(r"Depth=(\d+)", c)
            cur = (m.group(1)[2:], int(d.group(1)) if d else 1)
        elif h:
            cur = (h.group(1), int(h.group(2)))
        else:
            cur = ("-", 0)
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    cat = ("mfma" if "mfma" in op else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
           "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
           "wait/nop" if op in ("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio") else
           "branch" if op.startswith(("s_cbranch", "s_branch")) else "salu" if op.startswith("s_") else "other")
    counts[cur][cat] += 1
    if cat in ("valu", "salu"):
        counts[cur]["op:" + re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op)] += 1
for k in sorted(counts, key=lambda k: -sum(v for kk, v in counts[k].items() if not kk.startswith("op:"))):
    c = counts[k]
    tot = sum(v for kk, v in c.items() if not kk.startswith("op:"))
    if tot < thresh:
        continue
    print(f"loop {k[0]} depth {k[1]}: total {tot}", {kk: v for kk, v in c.items() if not kk.startswith("op:")})
    top = sorted(((v, kk[3:]) for kk, v in c.items() if kk.startswith("op:")), reverse=True)[:14]
    print("      ", " ".join(f"{kk}:{v}" for v, kk in top))
