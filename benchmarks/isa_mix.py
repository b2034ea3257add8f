"""Instruction mix of one kernel of the gfx950 assembly (hipcc -save-temps), split at basic-block labels and
s_barrier: python benchmarks/isa_mix.py <file.s> <mangled-kernel-prefix> [min_instrs]"""
import collections
import re
import sys

src = open(sys.argv[1]).read()
name = sys.argv[2]
thresh = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = src.index("\n" + name) + 1
end = src.index(".Lfunc_end", start)
body = src[start:end].split("\n")[1:]
seg, counts, labels = 0, collections.defaultdict(collections.Counter), {}
for ln in body:
    t = ln.strip()
    if not t or t.startswith(";"):
        continue
    if t.startswith(".LBB") or t.endswith(":"):
        seg += 1
        labels[seg] = t.split(":")[0]
        continue
    if t.startswith("."):
        continue
    op = t.split()[0]
    if op == "s_barrier":
        counts[seg]["BARRIER"] += 1
        seg += 1
        continue
    cat = ("mfma" if "mfma" in op else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
           "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
           "salu" if op.startswith("s_") else "other")
    counts[seg][cat] += 1
    if cat == "valu":
        counts[seg]["op:" + re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op)] += 1
for s in sorted(counts):
    c = counts[s]
    tot = sum(v for k, v in c.items() if not k.startswith("op:") and k != "BARRIER")
    if tot < thresh:
        continue
    top = sorted(((v, k[3:]) for k, v in c.items() if k.startswith("op:")), reverse=True)[:10]
    print(s, labels.get(s, ""), {k: v for k, v in c.items() if not k.startswith("op:")})
    print("      ", " ".join(f"{k}:{v}" for v, k in top))
