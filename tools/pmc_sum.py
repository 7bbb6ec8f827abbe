"""Summarise the counters of a rocprofv3 --pmc run (rocpd sqlite file): per kernel, grid size and counter the
mean value per launch. Usage: pmc_sum.py results.db [out.md]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, grid_size, counter_name, count(*), avg(value) from counters_collection "
                  "group by kernel_name, grid_size, counter_name order by kernel_name, grid_size desc").fetchall()
lines = ["| kernel | grid (threads) | counter | launches | mean per launch |", "|---|---|---|---|---|"]
for name, grid, ctr, n, v in rows:
    name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))[:80]
    lines.append("| %s | %d | %s | %d | %.1f |" % (name, grid, ctr, n, v))
text = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
print(text)
