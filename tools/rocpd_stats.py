"""Summarise a rocprofv3 rocpd sqlite file (kernel-trace) into the --stats style table: per kernel
calls, total, average, min, max duration (us) and share. Usage: rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(grid_x*grid_y*grid_z), max(workgroup_x), max(vgpr_count), max(lds_size) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total us | avg us | min us | max us | % | max grid | wg | vgpr | lds |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f | %d | %d | %d | %d |" % (
            short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7], r[8], r[9]))
    # the kNN kernel runs at several shapes in one bench (warm-up frames, the single-frame tail, the batch): its
    # launches are listed per grid as well, so the batch launch can be compared with bench.py's roofline.launch_ms
    per = db.execute("select name, grid_x*grid_y*grid_z, count(*), avg(duration), min(duration), max(duration) "
                     "from kernels where name like '%knn2_kernel%' group by name, grid_x*grid_y*grid_z "
                     "having avg(duration) > 1000000 order by 2 desc").fetchall()
    if per:
        lines += ["", "kNN kernel per launch shape (launches of 1 ms and more):", "",
                  "| kernel | grid (threads) | calls | avg us | min us | max us |", "|---|---|---|---|---|---|"]
        for r in per:
            lines.append("| %s | %d | %d | %.2f | %.2f | %.2f |" % (short(r[0]), r[1], r[2], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


main()
