"""BASELINE.json configs[2]: the frame-by-frame position_finder loop on 1 GPU, 1280x720 frames of a synthetic
flight against a synthetic orthomap (the reference's videos and map are not in its tree). Prints one JSON
line: frames/s of the whole loop (host Python included), fix rate and position error against the planted
truth, and how often each hot function ran (the reference's cadence, position_finder.py:232-240,275-287,316).
Usage: replay_c3.py [n_frames] [map_side]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from image_matching_amd import replay, synth
from image_matching_amd.match_finder import match_finder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
side = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
# m/s a fix may be away from the dead-reckoned position (position_finder.py:547-559). The reference's odometry signs
# (:415-418) belong to its camera mounting; on the synthetic frames every odometry tick drags the dead-reckoned position
# away, and with a flight-like gate (try 60) the next few fixes are then rejected. Default: gate open.
lowpass = float(sys.argv[3]) if len(sys.argv) > 3 else 1e6
map_px, frame_px = 0.5, 0.375
m = synth.synth_map(side, seed=1234)
flight = synth.synth_flight(m, n, map_pixel_size=map_px, frame_pixel_size=frame_px, seed=1)
mf = match_finder(detector=os.environ.get("REPLAY_DETECTOR", "sift"))   # "akaze": the detector the checked-in file runs
mf.draw_matches = False          # the drawMatchesKnn canvas is only published on request (position_finder.py:319-320)
main_map = replay.make_map(m, map_px)
poi = replay.DEFAULT_CONFIG["camera_poi"] / 180.0 * np.pi
height = frame_px * 1280 / (2 * np.tanh(poi / 2))        # so that find_pixel_size_by_height gives frame_px
g = replay.GeodeticConvert()
g.initialiseReference(main_map.main_points[0].lat, main_map.main_points[0].lon, 0)


map_roi = None
if os.environ.get("REPLAY_MAPDB"):          # every frame against the whole resident map instead of ROI crops
    from image_matching_amd.mapdb import MapDB
    t0 = time.perf_counter()
    db = MapDB.build(mf.ctx, m, tile=4096, margin=256, pixel_size=map_px)
    print("map database: %d descriptors in %.2f s" % (len(db), time.perf_counter() - t0), file=sys.stderr)
    map_roi = db.as_roi(mf.ctx, main_map)


def run(frames):
    pf = replay.PositionFinder(main_map, mf, {"height": height, "low_pass_coordinates": lowpass}, map_roi=map_roi)
    t0 = time.perf_counter()
    for img, tr in frames:
        lat, lon, _ = g.ned2Geodetic(north=-tr["cy"] * map_px, east=tr["cx"] * map_px, down=0)
        pf.photo_cb(img, imu=(0.0, 0.0, tr["yaw"]), gps=(lat, lon), stamp=tr["stamp"])
    mf.ctx.synchronize()
    return pf, time.perf_counter() - t0


run(flight[:12])                 # warm-up: arenas, pools
import gc
gc.collect()
gc.disable()                     # a generational collection (~35 ms here) inside the loop would read as frame time
pf, wall = run(flight)
gc.enable()
truth = {tr["stamp"]: tr for _, tr in flight}
err = [np.hypot(f["x_m"] - truth[f["stamp"]]["cx"] * map_px, f["y_m"] - truth[f["stamp"]]["cy"] * map_px) for f in pf.fixes]
if os.environ.get("REPLAY_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    run(flight[:60])
    pr.disable()
    pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(28)
print(json.dumps({
    "metric": "frames/sec, position_finder loop (configs[2])", "detector": mf.detector, "value": n / wall, "unit": "frames/s", "n_gpus": 1,
    "frames": n, "frame": "1280x720 u8", "map": "%dx%d synthetic, %.2f m/px" % (side, side, map_px),
    "ms_per_frame": wall / n * 1e3, "fixes": len(pf.fixes), "fix_rate": len(pf.fixes) / n,
    "position_error_m": {"median": float(np.median(err)) if err else None, "p90": float(np.percentile(err, 90)) if err else None,
                         "max": float(np.max(err)) if err else None, "map_pixel_m": map_px},
    "roi_source": "resident map database (whole map)" if map_roi is not None else "crop around the last fix, re-extracted every 12th frame",
    "calls": pf.calls, "odometry_ticks": len(pf.odometry), "swallowed_errors": len(pf.errors),
    "mean_keypoints": {"frame": float(np.mean([f["cadr_kp"] for f in pf.frames])), "roi": float(np.mean([f["roi_kp"] for f in pf.frames]))},
}))
