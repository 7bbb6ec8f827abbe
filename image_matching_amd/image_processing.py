"""The frame/map container the reference passes around (reference image_processing.py:16-111):
`img, kp, dp, pixel_size, cadr_scale, main_points` and find_kp_dp_scale, the caller of the hot path
(image_processing.py:109-111). Map loading from ~/copa5/map/<name>.tif + .@@@ corner file is kept for
completeness (Pillow instead of cv2.imread; image_processing.py:34-69)."""
import os
from dataclasses import dataclass
from decimal import Decimal

import numpy as np

from .geodetic_conv import GeodeticConvert


@dataclass
class img_point:
    pixel_y: int = 0
    pixel_x: int = 0
    lat: float = 0.0
    lon: float = 0.0


class image_processing():
    def __init__(self, filename=None, img=None):
        self.cuda = True
        self.main_points = []
        self.g_c = GeodeticConvert()
        self.img = None
        self.pixel_size = 0
        self.kp = None
        self.dp = None
        self.cadr_scale = 0.0
        if filename is not None:
            data_path = os.path.join(os.getenv("HOME", ""), "copa5", "map")
            path = os.path.join(data_path, filename + ".tif")
            if not os.path.exists(path):
                path = os.path.join(data_path, filename + ".TIF")
            from PIL import Image
            Image.MAX_IMAGE_PIXELS = None
            bgr = np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)[..., ::-1])  # imread order
            # the reference applies COLOR_RGB2GRAY to imread's BGR data (image_processing.py:42,54): OpenCV's
            # weights land on the swapped channels; same kernel as the frames' BGR2GRAY (imatch_cvt_gray)
            from .utils import default_context
            self.img = default_context().cvt_gray(bgr, rgb=True)
            with open(os.path.join(data_path, filename + ".@@@")) as f:
                for line in f.readlines()[2:]:
                    vals = [float(t) for t in line.split()]
                    if len(vals) >= 4:
                        self.main_points.append(img_point(vals[0], vals[1], vals[2], vals[3]))
        else:
            self.img = img

    def find_pixel_size(self):
        self.g_c.initialiseReference(self.main_points[0].lat, self.main_points[0].lon, 0)
        x_1, y_1, _ = self.g_c.geodetic2Ned(self.main_points[1].lat, self.main_points[1].lon, 0)
        x_2, y_2, _ = self.g_c.geodetic2Ned(self.main_points[3].lat, self.main_points[3].lon, 0)
        x = x_1 if abs(x_1) > abs(x_2) else x_2
        y = y_1 if abs(y_1) > abs(y_2) else y_2
        pixel_size_1 = Decimal(abs(x)) / Decimal(self.img.shape[0])
        pixel_size_2 = Decimal(abs(y)) / Decimal(self.img.shape[1])
        self.pixel_size = (Decimal(pixel_size_1) + Decimal(pixel_size_2)) / Decimal(2)
        return self.pixel_size

    def find_pixel_size_by_height(self, height, poi):
        # np.tanh, not tan, as in the reference (image_processing.py:93)
        x = Decimal(np.tanh(poi / 2) * 2 * height)
        self.pixel_size = x / Decimal(self.img.shape[1])

    def is_cuda_cv(self):
        return self.cuda

    def find_kp_dp_scale(self, match_finder):
        self.img, self.cadr_scale, self.pixel_size = match_finder.rescale_cadr(self.img, self.pixel_size)
        self.kp, self.dp, self.img = match_finder.find_kp_dp(self.img)
