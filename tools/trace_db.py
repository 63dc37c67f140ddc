"""Launch sequence of ONE call out of a rocprofv3 --kernel-trace result database (rocpd sqlite): the kernels between two
launches of an anchor kernel, with start offsets, durations and the idle gaps between them.
    python tools/trace_db.py <t_results.db> <anchor substring> [which occurrence, default: the last but one]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
    namecol = "display_name" if "display_name" in cols else "kernel_name"
    rows = list(cur.execute("select d.start, d.end, s.%s, d.grid_size_x, d.workgroup_size_x, d.group_segment_size from %s d join %s s on d.kernel_id = s.id "
                            "order by d.start" % (namecol, kd, ks)))
    anchor = sys.argv[2]
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(idx) < 2:
        raise SystemExit("anchor found %d times" % len(idx))
    k = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) - 2
    a, b = idx[k], idx[k + 1]
    t0 = rows[a][0]
    prev_end = None
    busy = 0
    for r in rows[a:b]:
        name = r[2].replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*", "", name)
        name = re.sub(r"^void ", "", name).replace("smh::", "")
        gap = (r[0] - prev_end) / 1e3 if prev_end is not None else 0.0
        print("%9.1f us  +%6.1f gap  %8.1f us  grid %8d x %4d lds %6d  %s" % ((r[0] - t0) / 1e3, gap, (r[1] - r[0]) / 1e3, r[3] // max(1, r[4]), r[4], r[5], name[:90]))
        busy += r[1] - r[0]
        prev_end = max(prev_end or 0, r[1])
    print("launches %d, span %.1f us, kernels busy %.1f us" % (b - a, (rows[b][0] - t0) / 1e3, busy / 1e3))


if __name__ == "__main__":
    main()
