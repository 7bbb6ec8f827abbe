"""Development: the kernels of one batched bench step (between two consecutive launches of the 512-query kNN kernel) from a
rocprofv3 rocpd database, with start offsets, durations and gaps. Usage: bench_step_timeline.py results.db [which] [out.md]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = db.execute("select name, start, end, grid_x*grid_y*grid_z from kernels order by start").fetchall()
knn = [i for i, r in enumerate(rows) if r[0].startswith("void knn2_kernel16<8") or r[0].startswith("knn2_kernel16<8")]
a, b = knn[which], knn[which + 1]
t0 = rows[a][1]
prev = t0
out = ["| start us | dur us | gap us | kernel |", "|---|---|---|---|"]
busy = 0
for name, s, e, g in rows[a:b + 1]:
    nm = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))[:50]
    out.append("| %.1f | %.1f | %.1f | %s |" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, nm))
    busy += e - s
    prev = max(prev, e)
out.append("")
out.append("step span %.1f us (kNN launch to kNN launch)" % ((rows[b][1] - t0) / 1e3))
text = "\n".join(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(text + "\n")
print(text)
