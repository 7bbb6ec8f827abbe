"""Print the kernel timeline of the last N kernels of a rocprofv3 rocpd sqlite file: start offset,
duration and gap to the previous kernel (us). Usage: rocpd_timeline.py results.db [n_last] [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return re.sub(r"^void ", "", name)[:60]


db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = db.execute("select name, start, end, grid_x*grid_y*grid_z from kernels order by start").fetchall()
rows = rows[-n:]
t0 = rows[0][1]
prev_end = t0
lines = ["| start us | dur us | gap us | grid | kernel |", "|---|---|---|---|---|"]
busy = 0
for name, s, e, g in rows:
    lines.append("| %.1f | %.2f | %.2f | %d | %s |" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, g, short(name)))
    busy += e - s
    prev_end = e
lines.append("")
lines.append("span %.1f us, kernel-busy %.1f us" % ((rows[-1][2] - t0) / 1e3, busy / 1e3))
text = "\n".join(lines)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(text + "\n")
print(text)
