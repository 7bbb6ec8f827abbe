"""Merge the FETCH_SIZE and WRITE_SIZE passes (tools/pmc_sum.py tables) into profiles/rN_pmc.md and pull the
dominant kNN launch's traffic into profiles/rN_pmc_knn.json (read by bench.py).
Usage: pmc_merge.py pmc_fetch.md pmc_write.md bench.json out_pmc.md out_pmc_knn.json"""
import json
import sys

HEAD = """# PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs of `bench.py --steps 2 --warmup 1 --no-cpu-baseline`)

FETCH_SIZE and WRITE_SIZE are in KB, mean per launch. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE tallies the 128-B requests of
16-B-per-lane streaming reads at 64 B on gfx950, so the corrected figure is 2 x FETCH_SIZE. The extremum scans read 4, 8 or 16 B per
lane (detect_stream_kernel, detect_streamv_kernel<2>, <4>); tools/fetch_calib.hip reads a known 2 GiB in exactly those patterns and
FETCH_SIZE reports half of it for all three (profiles/r3_fetch_calib.md), so the same factor applies to them. WRITE_SIZE is exact.
A 16384^2 f32 layer is 1048576 KB.

| kernel | grid (threads) | launches | FETCH_SIZE KB | 2 x FETCH MB | WRITE_SIZE KB |
|---|---|---|---|---|---|
"""
KEEP = ("detect_stream", "blur_march_kernel", "knn2_kernel")


def table(path):
    out = {}
    for line in open(path):
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) == 5 and c[1].isdigit():
            out[(c[0], int(c[1]))] = (int(c[3]), float(c[4]))
    return out


def main():
    f, w = table(sys.argv[1]), table(sys.argv[2])
    bench = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    rows = []
    for (k, g) in sorted(f, key=lambda kg: (kg[0], -kg[1])):
        if not k.startswith(KEEP) or g < 256 * 1024:
            continue
        fk = f[(k, g)][1]
        wk = w.get((k, g), (0, float("nan")))[1]
        cal = "%.1f" % (2 * fk / 1024)
        rows.append("| %s | %d | %d | %.0f | %s | %.0f |" % (k, g, f[(k, g)][0], fk, cal, wk))
    open(sys.argv[4], "w").write(HEAD + "\n".join(rows) + "\n")
    # the dominant kNN launch = the batch's launch seen most often among the large grids
    knn = [(kg, v) for kg, v in f.items() if kg[0].startswith("knn2_kernel") and kg[1] > 1024 * 1024]
    (k, g), (n, fk) = max(knn, key=lambda kv: kv[1][0])
    wk = w[(k, g)][1]
    json.dump({"kernel": k, "grid_threads": g, "fetch_kb": fk, "write_kb": wk,
               "traffic_bytes_per_launch": (2 * fk + wk) * 1024.0, "batch": bench["config"]["frames_per_step_per_gpu"],
               "workload": "%d frames (about %dk queries) x %d map rows" % (
                   bench["config"]["frames_per_step_per_gpu"],
                   round(bench["config"]["keypoints_per_frame"] * bench["config"]["frames_per_step_per_gpu"] / 1000),
                   bench["config"]["map_descriptors"]),
               "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes; fetch doubled per "
                         "MI355X_MICROARCH.md",
               "n_queries": bench["config"]["keypoints_per_frame"] * bench["config"]["frames_per_step_per_gpu"]},
              open(sys.argv[5], "w"), indent=1)


main()
