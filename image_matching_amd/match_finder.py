"""Drop-in for the reference's matching facade (image_processing/src/image_processing/match_finder.py):
the class `match_finder` and the dataclass `roi` with the reference's method names, positional
signatures and return arities, so position_finder.py / mezhcadr.py / image_processing.py run
unchanged on top of it. The cv2 calls of the hot path are replaced by libimatch.so (HIP, gfx950):

    find_kp_dp               CLAHE + SIFT detectAndCompute        (ref match_finder.py:199-221)
    find_matches             BF kNN k=2 + Lowe ratio              (ref match_finder.py:177-197)
    find_keypoints_transform robust homography + plausibility gate (ref match_finder.py:223-255)
    rescale_* / resize_by_scale  INTER_AREA resize                (ref match_finder.py:165-175,308-311)

The checked-in reference runs AKAZE in find_kp_dp (match_finder.py:210-211) with the SIFT call
commented out (match_finder.py:205-209); this build implements the SIFT call, whose six parameters
the constructor still reads (match_finder.py:29-34), as BASELINE.json's north_star names it.

There is no CPU fallback: without libimatch.so or without a GPU, construction raises."""
import math
from dataclasses import dataclass
from decimal import Decimal

import numpy as np

from . import draw
from ._lib import AkazeParams, SiftParams
from .cvtypes import DeviceDescriptors, FloatDescriptors, GoodMatches, KeyPoints
from .geodetic_conv import GeodeticConvert
from .params import get_param, get_param_or_default
from .utils import default_context, isConvex, line_intersection, resize_img


@dataclass
class roi:
    img: np.ndarray
    kp: tuple
    dp: np.ndarray
    pixel_size: None
    pixel_x_main_map: int = 0
    pixel_y_main_map: int = 0


def _features_of(ctx, dp):
    """Device handle of a descriptor matrix: the one attached by find_kp_dp, or an upload of a plain
    ndarray (what a caller holding cv2 descriptors would pass)."""
    f = getattr(dp, "features", None)
    if f is not None and f.ctx is ctx and f._h is not None:
        return f
    return ctx.features(np.asarray(dp, dtype=np.float32))


class match_finder():
    def __init__(self, ctx=None, detector=None):
        # "sift": the detector north_star names (the commented call of match_finder.py:205-209); "akaze": the call the
        # checked-in file makes (match_finder.py:210-211). Everything downstream is the same code.
        if detector is None:
            detector = get_param_or_default("detector")
        if detector not in ("sift", "akaze"):
            raise ValueError("detector must be 'sift' or 'akaze'")
        self.detector = detector
        self.akaze_threshold, self.akaze_octaves, self.akaze_layers = 0.0005, 4, 5   # match_finder.py:210-211
        # parameters (same keys as the reference, match_finder.py:29-38)
        self.optimal_size_x = get_param("image_size_sift")
        self.points_quality = get_param("points_quality_sift")
        self.nOctaveLayers = get_param("nOctaveLayers_sift")
        self.contrastThreshold = get_param("contrastThreshold_sift")
        self.edgeThreshold = get_param("edgeThreshold_sift")
        self.sigma = get_param("sigma_sift")
        self.angle_restriction_homography = get_param("angle_restriction_homography")
        self.low_scale_restriction_homography = get_param("low_scale_restriction_homography")
        self.high_scale_restriction_homography = get_param("high_scale_restriction_homography")
        self.camera_pitch_angle = get_param("camera_pitch_angle")
        self.ctx = ctx if ctx is not None else default_context()  # raises without a GPU
        self.cuda_enabled = self.is_cuda_cv()
        self.search_scale = 2
        self.roi_img = None
        self.percent_of_good_value = 1.0
        # extras of this build (not in the reference)
        self.draw_matches = True        # build the drawMatchesKnn canvas in find_matches
        self.homography_seed = 0        # sampling seed of the hypothesis generator
        self.homography_threshold = 0.1  # the reference's findHomography threshold (match_finder.py:233)

    # ---- scale / ROI bookkeeping (host Python, reference behaviour) ------------------------
    def find_scale(self, pixel_size_map, pixel_size_cadr):
        if pixel_size_map > pixel_size_cadr:
            return Decimal(pixel_size_cadr / pixel_size_map), True
        return Decimal(pixel_size_map / pixel_size_cadr), False

    @staticmethod
    def _crop(map_, x_min, x_max, y_min, y_max):
        x_min, y_min = max(x_min, 0), max(y_min, 0)
        x_max, y_max = min(x_max, map_.img.shape[1]), min(y_max, map_.img.shape[0])
        return roi(img=map_.img[y_min:y_max, x_min:x_max], kp=None, dp=None, pixel_size=map_.pixel_size,
                   pixel_x_main_map=x_min, pixel_y_main_map=y_min)

    def find_map_roi_by_coordinates(self, map_, cadr, lat, lon, search_scale):
        g_c = GeodeticConvert()
        g_c.initialiseReference(map_.main_points[0].lat, map_.main_points[0].lon, 0)
        y, x, z = g_c.geodetic2Ned(lat, lon, 0)
        pixel_x = int(Decimal(x) / map_.pixel_size)
        pixel_y = int(Decimal(-y) / map_.pixel_size)
        width = cadr.img.shape[1] * search_scale / cadr.cadr_scale
        height = cadr.img.shape[0] * search_scale / cadr.cadr_scale
        return self._crop(map_, int(pixel_x - width / 2), int(pixel_x + width / 2), int(pixel_y - height / 2),
                          int(pixel_y + height / 2))

    def roi_from_map(self, map_, rl_size_x, rl_size_y):
        map_shape_x = map_.img.shape[1]
        map_shape_y = map_.img.shape[0]
        _, cel_x = math.modf(map_shape_x / rl_size_x)
        _, cel_y = math.modf(map_shape_y / rl_size_y)
        step_x = rl_size_x + ((map_shape_x - rl_size_x * cel_x) / cel_x) if cel_x != 0 else map_shape_x
        # the reference uses cel_x in the numerator here (match_finder.py:98); kept
        step_y = rl_size_y + ((map_shape_y - rl_size_y * cel_x) / cel_y) if cel_y != 0 else map_shape_y
        borders = []
        for x in range(0, int(map_shape_x), int(step_x)):
            x_min = int(x)
            x_max = min(int(x + step_x + step_x * 0.1), map_shape_x)
            if x_max - x_min < step_x / 5:
                break
            for y in range(0, int(map_shape_y), int(step_y)):
                y_min = int(y)
                y_max = min(int(y + step_y + step_y * 0.1), map_shape_y)
                if y_max - y_min < step_y / 5:
                    break
                borders.append([x_min, x_max, y_min, y_max])
        return borders

    def create_roi_from_border(self, map_, border):
        x_min, x_max, y_min, y_max = border[0], border[1], border[2], border[3]
        # axes as in the reference (match_finder.py:125): rows are sliced with the x range
        img = map_.img[x_min:x_max, y_min:y_max]
        return roi(img=img, kp=None, dp=None, pixel_size=map_.pixel_size, pixel_x_main_map=x_min,
                   pixel_y_main_map=y_min)

    def roi_full_map(self, map_):
        return roi(map_.img, 0, 0, map_.pixel_size)

    def roi_from_last_xy(self, map_, x_meter, y_meter, cadr, search_scale, yaw):
        pixel_x = x_meter / float(map_.pixel_size)
        pixel_y = y_meter / float(map_.pixel_size)
        width = cadr.img.shape[1] * search_scale / cadr.cadr_scale
        height = cadr.img.shape[0] * search_scale / cadr.cadr_scale
        width = height = max(width, height)
        return self._crop(map_, int(pixel_x - width / 2), int(pixel_x + width / 2), int(pixel_y - height / 2),
                          int(pixel_y + height / 2))

    def rescale_for_optimal_sift(self, roi, cadr):
        optimal_scale_for_cadr = cadr.cadr_scale
        roi.img = resize_img(roi.img, optimal_scale_for_cadr)
        roi.pixel_size = roi.pixel_size / Decimal(optimal_scale_for_cadr)
        return roi, cadr

    def rescale_cadr(self, cadr, pixel_size):
        optimal_scale_for_cadr = self.optimal_size_x / cadr.shape[1]
        cadr = resize_img(cadr, optimal_scale_for_cadr)
        pixel_size = pixel_size / Decimal(optimal_scale_for_cadr)
        return cadr, optimal_scale_for_cadr, pixel_size

    def resize_by_scale(self, img, pixel_size, scale):
        img = resize_img(img, scale)
        pixel_size = Decimal(pixel_size) / Decimal(scale)
        return img, pixel_size

    # ---- the hot path ------------------------------------------------------------------------
    def sift_params(self, clahe=True):
        return SiftParams(self.nOctaveLayers, self.contrastThreshold, self.edgeThreshold, self.sigma, clahe)

    def find_kp_dp(self, img):
        """-> (keypoints, descriptors, CLAHE'd image), as the reference returns them."""
        if self.detector == "akaze":
            rec, desc, out = self.ctx.akaze_extract(img, AkazeParams(self.akaze_threshold, self.akaze_octaves,
                                                                     self.akaze_layers, clahe=True), image=True)
            return KeyPoints(rec), FloatDescriptors(desc, source=(self.ctx, self.ctx.akaze_last()[0])), out
        f = self.ctx.sift_extract(img, self.sift_params(clahe=True))
        kp = KeyPoints(f.download(descriptors=False)[0])
        return kp, DeviceDescriptors(f), f.image(img.shape[0], img.shape[1])

    def knn_match(self, query_dp, train_dp):
        """knnMatch(query, train, k=2) -> (idx int32 [Q,2], dist f32 [Q,2]); (-1, inf) where missing."""
        if not hasattr(query_dp, "features") and np.ndim(query_dp) == 2 and np.shape(query_dp)[1] != 128:
            # float descriptors (AKAZE's 64-float M-SURF); a train matrix made by find_kp_dp is uploaded once
            if isinstance(train_dp, FloatDescriptors) and len(train_dp) > 0:
                if train_dp.fset is None or train_dp.fset.ctx is not self.ctx or train_dp.fset._h is None:
                    train_dp.fset = self.ctx.fset(train_dp)
                # the frame extracted last (position_finder.py:202-203 then 232-240): its descriptors are still where
                # the extraction left them on the device
                src = getattr(query_dp, "source", None)
                if src is not None and src[0] is self.ctx and src[1] and train_dp.fset.d == 64 and \
                        self.ctx.akaze_last() == (src[1], len(query_dp)):
                    return self.ctx.knn2_fset_akaze(src[1], train_dp.fset)
                return self.ctx.knn2_f32(query_dp, train_dp.fset)
            return self.ctx.knn2_f32(query_dp, train_dp)
        fq, ft = _features_of(self.ctx, query_dp), _features_of(self.ctx, train_dp)
        if fq.integral and ft.integral:
            return self.ctx.knn2(fq, ft)      # SIFT's integer rows: int8 MFMA, exact
        # 128 floats per row that are not integers 0..255 (RootSIFT, normalised rows, SURF-128): the MFMA kernel would
        # round them; BFMatcher's f32 arithmetic is the float kernel's (match_finder.py:182-183)
        return self.ctx.knn2_f32(np.asarray(query_dp, dtype=np.float32), np.asarray(train_dp, dtype=np.float32))

    def find_matches(self, roi, cadr):  # roi, cadr
        idx, dist = self.knn_match(cadr.dp, roi.dp)
        if len(idx) and (idx[:, 1] < 0).any():
            # cv2 yields rows with fewer than two DMatch and the reference's `for m,n in matches`
            # raises on them (match_finder.py:192); the callers catch it (position_finder.py:222-224)
            raise ValueError("not enough values to unpack (expected 2, got %d)" % int((idx[0] >= 0).sum()))
        # Lowe ratio in the reference's float semantics: f32 distances widened to double (match_finder.py:193)
        keep = dist[:, 0].astype(np.float64) < self.points_quality * dist[:, 1].astype(np.float64)
        q = np.nonzero(keep)[0]
        good = GoodMatches(q, idx[q, 0], dist[q, 0])     # [[DMatch], ...] when read; see cvtypes.GoodMatches
        img3 = None
        if self.draw_matches:
            xy1 = cadr.kp.xy if hasattr(cadr.kp, "xy") else _xy(cadr.kp)
            xy2 = roi.kp.xy if hasattr(roi.kp, "xy") else _xy(roi.kp)
            if np.ndim(cadr.img) == 2 and np.ndim(roi.img) == 2:      # composed and drawn on the GPU
                img3 = self.ctx.draw_matches(cadr.img, np.asarray(xy1)[good.query_idx], roi.img,
                                             np.asarray(xy2)[good.train_idx])
            else:
                img3 = draw.matches_canvas(cadr.img, xy1, roi.img, xy2,
                                           list(zip(good.query_idx.tolist(), good.train_idx.tolist())))
        return good, img3

    def find_keypoints_transform(self, matches, roi, cadr, img_for_pub):
        img1 = cadr.img
        kp1, kp2 = cadr.kp, roi.kp
        xy1 = kp1.xy if hasattr(kp1, "xy") else _xy(kp1)
        xy2 = kp2.xy if hasattr(kp2, "xy") else _xy(kp2)
        if isinstance(matches, GoodMatches):
            qi, ti = matches.query_idx, matches.train_idx
        else:   # a plain list of [DMatch], as cv2 code would hand over
            qi = np.fromiter((m[0].queryIdx for m in matches), dtype=np.int64, count=len(matches))
            ti = np.fromiter((m[0].trainIdx for m in matches), dtype=np.int64, count=len(matches))
        src_pts = np.float32(xy1[qi]).reshape(-1, 1, 2)
        dst_pts = np.float32(xy2[ti]).reshape(-1, 1, 2)
        M, mask = self.ctx.homography(src_pts, dst_pts, self.homography_threshold, self.homography_seed)
        h, w = img1.shape
        pts = np.float32([[0, 0], [0, h - 1], [w - 1, h - 1], [w - 1, 0]]).reshape(-1, 1, 2)
        dst = perspective_transform(pts, M)  # raises on M None, as cv2 does; callers catch (position_finder.py:321)
        isConv, answer = isConvex(dst, img1.shape, self.angle_restriction_homography,
                                  self.low_scale_restriction_homography, self.high_scale_restriction_homography)
        x_center, y_center = line_intersection((dst[0][0], dst[2][0]), (dst[1][0], dst[3][0]))
        if img_for_pub is not None:
            dst[:, 0, 0] += img1.shape[1]
            xc, yc = line_intersection((dst[0][0], dst[2][0]), (dst[1][0], dst[3][0]))
            # cv2.circle(img, c, 10, 255, 5) + cv2.polylines(img, [np.int32(dst)], True, 255, 3) (match_finder.py:249-250;
            # the scalar colour 255 is Scalar(255, 0, 0)): drawn on the GPU
            q = np.int32(dst).reshape(4, 2)
            prims = [("circle", (int(xc), int(yc)), 10, 255, 5)]
            prims += [("line", tuple(int(v) for v in q[k]), tuple(int(v) for v in q[(k + 1) % 4]), 255, 3) for k in range(4)]
            img_for_pub = self.ctx.draw_overlay(img_for_pub, prims)
        if isConv is True:
            roll, pitch, yaw = self.get_angles_from_homography(M)
            return x_center, y_center, roll, pitch, yaw, M, img_for_pub, answer
        return None, None, None, None, None, None, img_for_pub, answer

    # ---- pixel -> geodetic and H -> angles (host math, reference behaviour) ------------------
    def solve_IK(self, x_center, y_center, height, roll, pitch, yaw, roi, map_):
        delta_pitch_pixels = -1 * height * np.sin(pitch + self.camera_pitch_angle) / float(roi.pixel_size)
        delta_roll_pixels = height * np.sin(roll) / float(roi.pixel_size)
        x_center = x_center + delta_pitch_pixels * np.sin(yaw) + delta_roll_pixels * np.cos(yaw)
        y_center = y_center - delta_pitch_pixels * np.cos(yaw) + delta_roll_pixels * np.sin(yaw)
        scale_roi_to_map = roi.pixel_size / map_.pixel_size
        scaled_x_center = Decimal(Decimal(float(x_center)) * scale_roi_to_map)
        scaled_y_center = Decimal(Decimal(float(y_center)) * scale_roi_to_map)
        y = Decimal(roi.pixel_y_main_map) + Decimal(scaled_y_center)
        x = Decimal(roi.pixel_x_main_map) + Decimal(scaled_x_center)
        x_meter = x * map_.pixel_size
        y_meter = y * map_.pixel_size
        g_c = GeodeticConvert()
        g_c.initialiseReference(map_.main_points[0].lat, map_.main_points[0].lon, 0)
        lat, lon, _ = g_c.ned2Geodetic(north=float(-y_meter), east=float(x_meter), down=0)
        return lat, lon, x, y, x_meter, y_meter

    def get_angles_from_homography(self, H):
        # yaw: rotation part of the upper-left 2x2 (polar decomposition through the SVD)
        u, _, vh = np.linalg.svd(H[0:2, 0:2])
        R = u @ vh
        yaw = np.arctan2(R[1, 0], R[0, 0])
        pitch = np.arctan2(-H[2, 1], H[1, 1])
        roll = np.arctan2(H[0, 2], H[2, 2])
        return roll, pitch, yaw

    # ---- image conditioning helpers -----------------------------------------------------------
    def normalize_images(self, roi, cadr):
        return self.ctx.clahe(roi, 30.0, (8, 8)), self.ctx.clahe(cadr, 30.0, (8, 8))

    def basicLinearTransform(self, img_original, alpha, beta):
        # cv2.convertScaleAbs: saturate_cast<uchar>(|alpha*x + beta|)
        v = np.abs(np.asarray(img_original, dtype=np.float64) * alpha + beta)
        return np.clip(np.rint(v), 0, 255).astype(np.uint8)

    def gammaCorrection(self, img_original, gamma):
        table = np.empty(256, np.uint8)
        for i in range(256):
            table[i] = np.clip(pow(i / 255.0, gamma) * 255.0, 0, 255)
        return table[np.asarray(img_original)]

    def is_cuda_cv(self):
        """The reference's GPU probe (match_finder.py:341-352); here: is a HIP device driving this facade."""
        try:
            return self.ctx.lib.imatch_device_count() > 0
        except Exception:
            return False


def _xy(kps):
    return np.array([k.pt for k in kps], dtype=np.float32).reshape(-1, 2)


def perspective_transform(pts, M):
    """cv2.perspectiveTransform for float32 [n,1,2] points and a 3x3 double matrix."""
    if M is None:
        raise ValueError("perspectiveTransform: the transformation matrix is empty")
    M = np.asarray(M, dtype=np.float64)
    p = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    w = p[:, 0] * M[2, 0] + p[:, 1] * M[2, 1] + M[2, 2]
    w = np.where(np.abs(w) > np.finfo(np.float64).eps, 1.0 / np.where(w == 0, 1, w), 0.0)
    x = (p[:, 0] * M[0, 0] + p[:, 1] * M[0, 1] + M[0, 2]) * w
    y = (p[:, 0] * M[1, 0] + p[:, 1] * M[1, 1] + M[1, 2]) * w
    return np.stack([x, y], axis=1).astype(np.float32).reshape(-1, 1, 2)
