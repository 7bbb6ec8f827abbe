"""ROS-free counterpart of the reference's per-frame localisation loop, for BASELINE.json configs[2]
("flight replay at 1280x720, frame-by-frame position_finder loop on 1 GPU") and SURVEY.md 8(a) row a13.

It follows position_finder.py method by method -- photo_cb (:159-224), find_pose (:226-290), pose_from_roi
(:292-405), compare_cadrs (:407-443), low_pass_pose (:547-559) -- with the same state variables, the same
cadence (frame features every frame; ROI features re-extracted when roi_iterator > 10, i.e. every 12th
frame; one find_matches per frame plus one per odometry tick; find_keypoints_transform iff len(good) > 10)
and the same gates, on top of the drop-in `match_finder`. What is left out is the ROS plumbing: messages
become arguments (image, imu angles, gps, stamp), publishers become the `fixes` / `odometry` lists, the
debug images and the wind-measurement action are not built, and `time()` is the FRAME's stamp, so that a
replay does not depend on how fast it is processed.

Known reference defects are kept where they shape behaviour: the first odometry tick raises on
old_cadr = None and is swallowed (:301-309); `compare_cadrs` returns None when no transform is found, so the
caller's unpack raises and is swallowed too (:303-308, :442-443)."""
from decimal import Decimal

import numpy as np

from .geodetic_conv import GeodeticConvert
from .image_processing import image_processing, img_point
from .match_finder import match_finder

# caller-side parameters (position_finder.py:92-114); the flight YAML is not in the reference's tree, so these
# are working values for the synthetic flights, meant to be overridden
DEFAULT_CONFIG = {
    "camera_poi": 60.0,                             # degrees; photo_cb turns it into radians (:92)
    "search_scale_for_roi_by_gps": 2.0,
    "search_scale_for_roi_by_detection": 2.0,
    "search_scale_for_roi_by_rolling_window": 4.0,
    "count_of_pictures_for_odometry": 1.0,          # seconds between odometry ticks (:300)
    "low_pass_speed": 60.0,
    "low_pass_coordinates": 60.0,
    "use_gps": True,
    "use_baro": False,                              # True: the height follows the barometer relative to its first value (:595-603)
    "height": 500.0,
    "link_window_upscale": 2.0,
    "unlink_time_for_upscale": 5.0,
}


def make_map(img, pixel_size, lat0=55.0, lon0=37.0):
    """The map container position_finder builds from <name>.tif + .@@@ (image_processing.py:34-69): image,
    four geo-referenced corners (pixel (0,0) first) and the pixel size find_pixel_size derives from them."""
    m = image_processing(img=img)
    g = GeodeticConvert()
    g.initialiseReference(lat0, lon0, 0)
    h, w = img.shape
    for py, px in ((0, 0), (0, w), (h, w), (h, 0)):
        lat, lon, _ = g.ned2Geodetic(north=-py * pixel_size, east=px * pixel_size, down=0)
        m.main_points.append(img_point(py, px, lat, lon))
    m.pixel_size = Decimal(repr(pixel_size))
    return m


class PositionFinder:
    def __init__(self, main_map, matcher=None, config=None, map_roi=None):
        """map_roi: the whole map as one resident `roi` (mapdb.MapDB.as_roi). When given, find_pose matches every frame
        against it instead of cropping and re-extracting ROIs (no counterpart in the reference, whose ROI features are
        re-extracted every 12th frame)."""
        self.map_roi = map_roi
        cfg = dict(DEFAULT_CONFIG)
        cfg.update(config or {})
        self.main_map = main_map
        self.map_pixel_size = main_map.pixel_size
        self.matcher = matcher if matcher is not None else match_finder()
        # state, as position_finder.py:47-89
        self.first_cadr = True
        self.imu_roll = self.imu_pitch = self.imu_yaw = 0.0
        self.lat_gps = self.lon_gps = 0.0
        self.last_yaw = 0.0
        self.old_cadr = None
        self.now = 0.0
        self.time_between_cadrs = 0.0
        self.time_between_link = 0.0
        self.first_link = False
        self.upscale_link = False
        self.x_meter = 0
        self.y_meter = 0
        self.last_x_from_link = self.last_y_from_link = 0.0
        self.low_pass_time = 0.0
        self.rois = []
        self.roi_iterator = 100
        self.roi_after_link = None
        self.first_rolling = True
        self.pose_from_privyazka = False
        self.poi = cfg["camera_poi"] / 180.0 * np.pi
        self.search_scale_for_roi_by_gps = cfg["search_scale_for_roi_by_gps"]
        self.search_scale_for_roi_by_detection = cfg["search_scale_for_roi_by_detection"]
        self.search_scale_for_roi_by_rolling_window = cfg["search_scale_for_roi_by_rolling_window"]
        self.count_of_pictures_for_odometry = cfg["count_of_pictures_for_odometry"]
        self.low_pass_speed = cfg["low_pass_speed"]
        self.low_pass_coordinates = cfg["low_pass_coordinates"]
        self.use_gps = cfg["use_gps"]
        self.height = cfg["height"]
        self.use_baro = cfg["use_baro"]
        self.height_init = not self.use_baro         # position_finder.py:118-125
        self.start_height = 0.0
        self.link_window_upscale = cfg["link_window_upscale"]
        self.unlink_time_for_upscale = cfg["unlink_time_for_upscale"]
        # what the node would publish / log
        self.fixes = []       # /coordinates_by_img: dict(stamp, lat, lon, x_m, y_m)
        self.odometry = []    # /odom_by_img: dict(stamp, north_speed, east_speed, yaw_speed, from_privyazka)
        self.frames = []      # one record per photo_cb: counts and the gate's answer (log_keypoints, :393-397)
        self.errors = []      # what the reference's try/except blocks log and swallow
        self.calls = {"frame_extract": 0, "roi_extract": 0, "find_matches": 0, "find_keypoints_transform": 0}

    # ---- photo_cb (position_finder.py:159-224) ---------------------------------------------------------
    def baro_cb(self, value):
        """position_finder.py:595-603 for a replay (realtime False): height = barometer - its first value."""
        if self.height_init is False:
            self.height_init = True
            self.start_height = value
        self.height = value - self.start_height

    def photo_cb(self, image, imu=(0.0, 0.0, 0.0), gps=None, stamp=None, baro=None):
        if baro is not None and self.use_baro is True:
            self.baro_cb(baro)
        if self.use_baro is True and self.height_init is False:      # no height yet: the frame is dropped (:163)
            return None
        self.now = float(stamp) if stamp is not None else self.now + 0.2
        self._frame = {"stamp": self.now, "good": 0, "roi_kp": 0, "cadr_kp": 0, "from_privyazka": False, "answer": None}
        self.frames.append(self._frame)
        try:
            self.imu_roll, self.imu_pitch, self.imu_yaw = imu
            if gps is not None:
                self.lat_gps, self.lon_gps = gps
            cadr = image_processing(img=image)
            cadr.find_pixel_size_by_height(self.height, self.poi)
            scale, map_pixel_bigger = self.matcher.find_scale(self.map_pixel_size, cadr.pixel_size)
            cadr.img, cadr.pixel_size = self.matcher.resize_by_scale(cadr.img, cadr.pixel_size, scale)
            cadr.find_kp_dp_scale(self.matcher)
            self.calls["frame_extract"] += 1
            answer = self.find_pose(cadr)
            # upscale the search window if the link is lost for N seconds (:205-216)
            if answer is True and self.first_link is False:
                self.first_link = True
                self.time_between_link = self.now
            if answer is True and self.first_link is True:
                self.time_between_link = self.now
                self.upscale_link = False
            elif self.first_link is True:
                if self.now - self.time_between_link > self.unlink_time_for_upscale:
                    self.upscale_link = True
            return answer
        except Exception as e:   # the reference logs and goes on (:222-224)
            self.errors.append((self.now, "photo_cb", repr(e)))
            return None

    def _extract_roi(self, roi, cadr):
        roi, cadr = self.matcher.rescale_for_optimal_sift(roi, cadr)
        roi.kp, roi.dp, roi.img = self.matcher.find_kp_dp(roi.img)
        self.calls["roi_extract"] += 1
        return roi

    # ---- find_pose (position_finder.py:226-290) --------------------------------------------------------
    def find_pose(self, cadr):
        if self.map_roi is not None:
            return self.pose_from_roi(self.map_roi, cadr)
        if self.first_cadr is True:
            if self.use_gps is True:
                if self.roi_iterator > 10:
                    self.roi_after_link = self.matcher.find_map_roi_by_coordinates(
                        self.main_map, cadr, self.lat_gps, self.lon_gps, self.search_scale_for_roi_by_gps)
                    if self.roi_after_link.img.shape[0] == 0:
                        return None
                    self.roi_iterator = 0
                    self.roi_after_link = self._extract_roi(self.roi_after_link, cadr)
                else:
                    self.roi_iterator += 1
                if self.roi_after_link.img.shape[0] != 0:
                    return self.pose_from_roi(self.roi_after_link, cadr)
                return None
            # no GPS: tile the whole map once, then try every tile (:247-272)
            if self.first_rolling is True:
                win = (float(cadr.pixel_size) / float(self.main_map.pixel_size)) * cadr.img.shape[1] * \
                    self.search_scale_for_roi_by_rolling_window
                for border in self.matcher.roi_from_map(self.main_map, win, win):
                    roi = self.matcher.create_roi_from_border(self.main_map, border)
                    if roi.img.shape[0] > 0 and roi.img.shape[1] > 0:
                        roi = self._extract_roi(roi, cadr)
                        if len(roi.kp) > 0:
                            self.rois.append(roi)
                self.first_rolling = False
                return None
            for roi in self.rois:
                if self.pose_from_roi(roi, cadr) is True:
                    return True
            return False
        # tracking: ROI around the last position (:273-290)
        if self.roi_iterator > 10:
            scale = self.search_scale_for_roi_by_detection
            if self.upscale_link is not False:
                scale = scale * self.link_window_upscale
            self.roi_after_link = self.matcher.roi_from_last_xy(self.main_map, float(self.x_meter), float(self.y_meter),
                                                                cadr, scale, self.imu_yaw)
            self.roi_iterator = 0
            self.roi_after_link = self._extract_roi(self.roi_after_link, cadr)
        else:
            self.roi_iterator += 1
        return self.pose_from_roi(self.roi_after_link, cadr)

    # ---- pose_from_roi (position_finder.py:292-405) ----------------------------------------------------
    def pose_from_roi(self, roi, cadr):
        good, img_for_pub = self.matcher.find_matches(roi, cadr)
        self.calls["find_matches"] += 1
        speed_limit = False
        if self.now - self.time_between_cadrs > self.count_of_pictures_for_odometry:
            try:
                north_speed, east_speed, speed_limit, yaw_speed = self.compare_cadrs(cadr, self.old_cadr)
            except Exception as e:
                self.errors.append((self.now, "between cadrs", repr(e)))
            self.old_cadr = cadr
        answer = "transform exception"
        x_center = None
        yaw = 0.0
        if len(good) > 10:
            try:
                self.calls["find_keypoints_transform"] += 1
                x_center, y_center, roll, pitch, yaw, M, img_tf, answer = self.matcher.find_keypoints_transform(
                    good, roi, cadr, img_for_pub)
            except Exception as e:
                x_center = None
                self.errors.append((self.now, "find transform", repr(e)))
        else:
            answer = "not enough keypoints to transform"
        self.pose_from_privyazka = False
        if x_center is not None and speed_limit is False:
            lat_zero, lon_zero, _, _, x_meter, y_meter = self.matcher.solve_IK(x_center, y_center, self.height, 0, 0, yaw,
                                                                             roi, self.main_map)
            lat, lon, _, _, x_inc, y_inc = self.matcher.solve_IK(x_center, y_center, self.height, 0.0, 0.0, self.imu_yaw,
                                                                 roi, self.main_map)
            self.last_yaw = yaw
            if self.first_cadr is True:
                self.first_cadr = False
                self.x_meter = float(x_inc)
                self.y_meter = float(y_inc)
            if self.low_pass_pose(float(x_inc), float(y_inc)) is True:
                self.pose_from_privyazka = True
                self.x_meter = float(x_inc)
                self.y_meter = float(y_inc)
                self.last_x_from_link = float(x_inc)
                self.last_y_from_link = float(y_inc)
                g_c = GeodeticConvert()
                g_c.initialiseReference(self.main_map.main_points[0].lat, self.main_map.main_points[0].lon, 0)
                lat, lon, _ = g_c.ned2Geodetic(north=float(-y_inc), east=float(x_inc), down=0)
                self.fixes.append({"stamp": self.now, "lat": lat, "lon": lon, "x_m": float(x_inc), "y_m": float(y_inc),
                                   "yaw": float(yaw)})
        self._frame.update(good=len(good), roi_kp=len(roi.kp), cadr_kp=len(cadr.kp),
                           from_privyazka=self.pose_from_privyazka, answer=answer)
        return x_center is not None

    # ---- compare_cadrs (position_finder.py:407-443) ----------------------------------------------------
    def compare_cadrs(self, cadr, cadr_old):
        good, _ = self.matcher.find_matches(cadr, cadr_old)
        self.calls["find_matches"] += 1
        self.calls["find_keypoints_transform"] += 1
        x_center, y_center, roll, pitch, yaw_cadr, M, img, answer = self.matcher.find_keypoints_transform(
            good, cadr, cadr_old, None)
        if x_center is not None:
            delta_x = -1 * (x_center - cadr.img.shape[1] / 2) * float(cadr.pixel_size)
            delta_y = (y_center - cadr.img.shape[0] / 2) * float(cadr.pixel_size)
            x_trans = delta_y * np.sin(self.imu_yaw) - delta_x * np.cos(self.imu_yaw)
            y_trans = delta_y * np.cos(self.imu_yaw) - delta_x * np.sin(self.imu_yaw)
            delta_time = self.now - self.time_between_cadrs
            self.time_between_cadrs = self.now
            if delta_time < 4.0 and abs(yaw_cadr) < 1.0:
                north_speed = 1.5 * (y_trans / delta_time)
                east_speed = 1.5 * (x_trans / delta_time)
                speed_limit = False
                if abs(north_speed) > self.low_pass_speed or abs(east_speed) > self.low_pass_speed:
                    north_speed = 0
                    east_speed = 0
                    speed_limit = True
                yaw_speed = -1 * yaw_cadr / delta_time
                self.odometry.append({"stamp": self.now, "north_speed": north_speed, "east_speed": east_speed,
                                      "yaw_speed": yaw_speed, "from_privyazka": self.pose_from_privyazka})
                self.last_yaw -= yaw_cadr
                self.x_meter += east_speed * delta_time
                self.y_meter -= north_speed * delta_time
                return north_speed, east_speed, speed_limit, yaw_speed
        return None   # the caller's 4-way unpack raises, as in the reference

    # ---- low_pass_pose (position_finder.py:547-559) ----------------------------------------------------
    def low_pass_pose(self, x, y):
        delta_time = self.now - self.low_pass_time
        if delta_time <= 0:      # the reference divides wall-clock differences, never 0; a replay's first stamp can be
            delta_time = 1e-9
        vx = abs(self.x_meter - x) / delta_time
        vy = abs(self.y_meter - y) / delta_time
        if vx > self.low_pass_coordinates or vy > self.low_pass_coordinates:
            return False
        self.low_pass_time = self.now
        return True


# ---- replay feeder: a frames directory + the telemetry CSV -----------------------------------------
def parse_telemetry_csv(path):
    """The flight log the reference replays beside a video (csv_data_pkg/src/csv_data_pkg/publish_csv.py:26-40; sample
    csv_data_pkg/map/logs/log_1.csv): `;`-separated, one header row `time;lat;lon;alt;roll;pitch;head;ub;nsat`, time as
    H:MM:SS.ss. -> float64 [n, 9] = seconds (h*3600 + m*60 + s, as __parse_time_to_sec), lat, lon, alt, roll, pitch, head
    (degrees), ub, nsat."""
    import csv
    rows = []
    with open(path, newline="") as fh:
        for i, line in enumerate(csv.reader(fh, delimiter=";", quotechar="|")):
            if i == 0 or not line:
                continue
            h, m, sec = (float(v) for v in line[0].split(":"))
            rows.append([h * 3600 + m * 60 + sec] + [float(v) for v in line[1:9]])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 9)


class Flight:
    """A recorded flight as the reference's replay nodes feed it to position_finder: the frames of a video (here: a
    directory of image files, one per video frame, in name order) published at photo_publisher's cadence, each with the
    telemetry row that publish_csv would have sent last.

    photo_publisher.py:107-132 reads one video frame per loop at `rate = fps * rate_multiplier` Hz and publishes the frame
    of every loop in which its counter has passed rate / 5 (the counter then starts again): every sixth video frame at 25 fps,
    4.2 frames per second of replay, gray (cvtColor BGR2GRAY). publish_csv.py:41-48 sends row i at (t_i - t_0) / rate_multiplier seconds after
    the start, angles turned into radians (:53-57), altitude also as the barometer value (:75-76). Both clocks scale
    with rate_multiplier, so in VIDEO time frame k sits at k / fps and row i at t_i - t_0, which is what `messages`
    works in; a frame carries the last row at or before its time (the subscribers keep the latest message)."""

    def __init__(self, frame_paths, telemetry=None, fps=25.0, rate_multiplier=1.0, ctx=None):
        self.paths = list(frame_paths)
        self.telemetry = np.zeros((0, 9)) if telemetry is None else np.asarray(telemetry, dtype=np.float64).reshape(-1, 9)
        self.fps = int(fps)                      # int(cap.get(CAP_PROP_FPS)), photo_publisher.py:80
        self.rate = self.fps * rate_multiplier   # :81
        self.ctx = ctx

    @classmethod
    def from_dir(cls, frames_dir, telemetry_csv=None, fps=25.0, rate_multiplier=1.0, ctx=None,
                 extensions=(".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".pgm", ".npy")):
        import os
        names = sorted(n for n in os.listdir(frames_dir) if n.lower().endswith(extensions))
        if telemetry_csv is None:               # photo_publisher looks for <video>.csv beside the video (:185-193)
            cand = frames_dir.rstrip("/") + ".csv"
            telemetry_csv = cand if os.path.exists(cand) else None
        tel = parse_telemetry_csv(telemetry_csv) if telemetry_csv is not None else None
        return cls([os.path.join(frames_dir, n) for n in names], tel, fps, rate_multiplier, ctx)

    def published_indices(self):
        """Indices of the video frames photo_publisher publishes (its counter logic, :109-132)."""
        out, iterator = [], 0
        for k in range(len(self.paths)):
            if iterator > self.rate / 5:
                out.append(k)
                iterator = 0
            iterator += 1
        return out

    def _gray(self, path):
        if path.lower().endswith(".npy"):
            img = np.load(path)
        else:
            from PIL import Image
            with Image.open(path) as im:
                img = np.asarray(im if im.mode in ("L", "RGB") else im.convert("RGB"))
        if img.ndim == 2:
            return np.ascontiguousarray(img, dtype=np.uint8)
        # cv2.imread / VideoCapture hand out BGR and the node converts BGR2GRAY (:112); a file decoded as RGB is the same
        # picture with the channels the other way round: the RGB form of the conversion gives the same gray values
        from .utils import default_context
        ctx = self.ctx if self.ctx is not None else default_context()
        return ctx.cvt_gray(np.ascontiguousarray(img[:, :, :3], dtype=np.uint8), rgb=True)

    def messages(self):
        """Yields dict(stamp, image, imu=(roll, pitch, yaw) in radians, gps=(lat, lon) or None, baro or None, nsat, index)
        for every published frame, in order."""
        tel = self.telemetry
        t_rel = tel[:, 0] - tel[0, 0] if len(tel) else np.zeros(0)
        for k in self.published_indices():
            stamp = k / float(self.fps)
            row = None
            if len(tel):
                j = int(np.searchsorted(t_rel, stamp, side="right")) - 1
                row = tel[j] if j >= 0 else None
            msg = {"stamp": stamp, "index": k, "image": self._gray(self.paths[k]), "imu": (0.0, 0.0, 0.0), "gps": None,
                   "baro": None, "nsat": None}
            if row is not None:
                msg["imu"] = tuple(float(v) / 180.0 * np.pi for v in row[4:7])
                msg["gps"] = (float(row[1]), float(row[2]))
                msg["baro"] = float(row[3])
                msg["nsat"] = int(row[8])
            yield msg

    def run(self, position_finder, limit=None):
        """Feed the flight to a PositionFinder (its photo_cb per published frame); returns the answers."""
        out = []
        for n, m in enumerate(self.messages()):
            if limit is not None and n >= limit:
                break
            out.append(position_finder.photo_cb(m["image"], imu=m["imu"], gps=m["gps"], stamp=m["stamp"], baro=m["baro"]))
        return out
