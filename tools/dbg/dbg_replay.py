import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from image_matching_amd import replay, synth
from image_matching_amd.match_finder import match_finder
m = synth.synth_map(4096, seed=99)
flight = synth.synth_flight(m, 26, map_pixel_size=0.5, frame_pixel_size=0.375, seed=3)
mf = match_finder(); mf.draw_matches = False
main_map = replay.make_map(m, 0.5)
poi = replay.DEFAULT_CONFIG["camera_poi"] / 180.0 * np.pi
pf = replay.PositionFinder(main_map, mf, {"height": 0.375 * 1280 / (2 * np.tanh(poi / 2))})
g = replay.GeodeticConvert(); g.initialiseReference(main_map.main_points[0].lat, main_map.main_points[0].lon, 0)
for img, tr in flight:
    lat, lon, _ = g.ned2Geodetic(north=-tr["cy"] * 0.5, east=tr["cx"] * 0.5, down=0)
    nf = len(pf.fixes)
    xm0, ym0 = pf.x_meter, pf.y_meter
    pf.photo_cb(img, imu=(0.0, 0.0, tr["yaw"]), gps=(lat, lon), stamp=tr["stamp"])
    fr = pf.frames[-1]
    fx = pf.fixes[-1] if len(pf.fixes) > nf else None
    print("t=%.1f good=%d roi_kp=%d cadr_kp=%d %s truth=(%.2f,%.2f) state_before=(%.2f,%.2f) fix=%s" % (
        tr["stamp"], fr["good"], fr["roi_kp"], fr["cadr_kp"], fr["answer"], tr["cx"]*0.5, tr["cy"]*0.5, float(xm0), float(ym0),
        None if fx is None else "(%.2f,%.2f) err=(%.2f,%.2f)" % (fx["x_m"], fx["y_m"], fx["x_m"]-tr["cx"]*0.5, fx["y_m"]-tr["cy"]*0.5)))
print(pf.odometry[:4])
print(pf.errors[:5])
