"""Per-kernel breakdown of the captured decode step from a rocprofv3 database:
   rocprofv3 --kernel-trace --stats -d DIR -o dec -- python benchmarks/decode_micro.py 64
   python benchmarks/decode_profile.py DIR/dec_results.db [n_last_kernels]"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
    tail = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 6000):]
    agg = collections.defaultdict(lambda: [0, 0])
    for n, s, e, g, w in tail:
        m = re.search(r"(\w+)(<[^>]*>)?\(", n.replace("(anonymous namespace)::", ""))
        k = f"{m.group(1) if m else n[:40]} wgs={g // max(w, 1)}"
        agg[k][0] += 1
        agg[k][1] += e - s
    busy = sum(v[1] for v in agg.values())
    print(f"kernels {len(tail)}  busy {busy / 1e6:.2f} ms  span {(tail[-1][2] - tail[0][1]) / 1e6:.2f} ms")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"{k:48s} n={v[0]:6d} avg={v[1] / v[0] / 1e3:7.2f} us  share={100 * v[1] / busy:5.1f} %")


if __name__ == "__main__":
    main()
