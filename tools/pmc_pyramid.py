"""HBM-side traffic of the scale-space stage of ONE map extraction, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
over `tools/extract_profile.py <side> <reps>` summarised by tools/pmc_sum.py: sum over every blur / scan / small-octave
launch of 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes; calibrated for the scans'
access widths in profiles/r3_fetch_calib.md), divided by the number of extractions.
Usage: pmc_pyramid.py pmc_fetch.md pmc_write.md side reps out.json [out.md]"""
import json
import sys

KEEP = ("blur_march_kernel", "detect_stream", "pyramid_tail", "blur3_tile", "upsample2", "decimate")


def table(path):
    out = {}
    for line in open(path):
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) == 5 and c[1].isdigit():
            out[(c[0], int(c[1]))] = (int(c[3]), float(c[4]))
    return out


f, w = table(sys.argv[1]), table(sys.argv[2])
side, reps = int(sys.argv[3]), int(sys.argv[4])
rows, total_f, total_w = [], 0.0, 0.0
for (k, g) in sorted(f, key=lambda kg: (kg[0], -kg[1])):
    if not k.startswith(KEEP):
        continue
    n, fk = f[(k, g)]
    wk = w.get((k, g), (0, 0.0))[1]
    total_f += 2 * fk * n
    total_w += wk * n
    rows.append("| %s | %d | %d | %.0f | %.1f | %.0f |" % (k, g, n, fk, 2 * fk / 1024, wk))
per = (total_f + total_w) * 1024.0 / reps
algo_moved = (64.0 * 16.0 / 3.0 + 17.0) * side * side
json.dump({"traffic_bytes_per_extraction": per, "read_bytes": total_f * 1024.0 / reps, "written_bytes": total_w * 1024.0 / reps,
           "map_side": side, "bytes_the_kernels_must_move": algo_moved, "ratio": per / algo_moved,
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over tools/extract_profile.py %d %d; "
                     "2 x FETCH + WRITE summed over every scale-space launch, per extraction" % (side, reps)},
          open(sys.argv[5], "w"), indent=1)
if len(sys.argv) > 6:
    head = ("# Scale-space stage of the %d x %d map extraction: HBM-side traffic per launch (rocprofv3 --pmc, %d extractions)\n\n"
            "Per extraction: read %.2f GB (2 x FETCH_SIZE), written %.2f GB, together %.2f GB = %.3f x the %.2f GB the kernels must move\n"
            "(each blur reads and writes a layer once, the scan reads six layers, the base blur reads the u8 image).\n\n"
            "| kernel | grid (threads) | launches | FETCH_SIZE KB | 2 x FETCH MB | WRITE_SIZE KB |\n|---|---|---|---|---|---|\n"
            % (side, side, reps, total_f * 1024 / reps / 1e9, total_w * 1024 / reps / 1e9, per / 1e9, per / algo_moved, algo_moved / 1e9))
    open(sys.argv[6], "w").write(head + "\n".join(rows) + "\n")
print(json.dumps({"traffic_GB": per / 1e9, "ratio": per / algo_moved}))
