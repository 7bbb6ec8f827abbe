"""Camera pre-step: the ROS-free counterpart of the capture node's `undistort` method
(reference csi_camera.py:138-167, identical in photo_publisher.py:143-172) and of the gray conversion the consumers
apply to the published frame (position_finder.py:168-171).

`undistort(img, balance, dim2, dim3, scale)` keeps the reference's signature, calibration constants and crop
arithmetic; the OpenCV calls inside it become libimatch entry points (include/imatch.h, "Camera pre-step"): the
rectification maps are built once per calibration and kept in HBM, and only the cropped window is computed.
"""
import numpy as np

from ._lib import Context, load_library, _ptr, c_f64p

# calibration of the reference's camera (csi_camera.py:142-144)
DIM = (1920, 1080)
K = np.array([[980.112887385314, 0.0, 941.3416228048669], [0.0, 984.9820720751916, 539.6096400309403], [0.0, 0.0, 1.0]])
D = np.array([[-0.027361240466794334], [0.02977144485814498], [-0.06669059605260248], [0.038761129156165954]])


def estimate_new_camera_matrix(K_, D_, size, balance=0.0):
    """cv2.fisheye.estimateNewCameraMatrixForUndistortRectify(K, D, size, np.eye(3), balance=balance)."""
    lib = load_library()
    k = np.ascontiguousarray(K_, np.float64).reshape(9)
    d = np.ascontiguousarray(D_, np.float64).reshape(4)
    out = np.empty(9, np.float64)
    rc = lib.imatch_fisheye_new_camera_matrix(_ptr(k, c_f64p), _ptr(d, c_f64p), int(size[0]), int(size[1]), float(balance),
                                              _ptr(out, c_f64p))
    if rc:
        raise ValueError("bad camera matrix arguments")
    return out.reshape(3, 3)


class Camera:
    def __init__(self, ctx=None, device=0, K_=K, D_=D, DIM_=DIM):
        self.ctx = ctx if ctx is not None else Context(device)
        self.K, self.D, self.DIM = np.array(K_, np.float64), np.array(D_, np.float64), tuple(DIM_)

    def undistort(self, img, balance=0.0, dim2=None, dim3=None, scale=0.5, to_gray=False):
        """csi_camera.undistort (csi_camera.py:138-167). to_gray=True also applies position_finder's
        cvtColor(BGR2GRAY) (position_finder.py:171) in the same kernel."""
        dim1 = img.shape[:2][::-1]
        assert dim1[0] / dim1[1] == self.DIM[0] / self.DIM[1], \
            "Image to undistort needs to have same aspect ratio as the ones used in calibration"
        if not dim2:
            dim2 = dim1
        if not dim3:
            dim3 = dim1
        if tuple(dim3) != tuple(dim1):
            raise NotImplementedError("dim3 != input size: the reference never passes it (csi_camera.py:184)")
        scaled_K = self.K * dim1[0] / self.DIM[0]
        scaled_K[2][2] = 1.0
        new_K = estimate_new_camera_matrix(scaled_K, self.D, dim2, balance)
        start_dim_0 = int((dim1[0] * (1 - scale)) / 2)
        start_dim_1 = int((dim1[1] * (1 - scale)) / 2)
        target_dim_0 = int(dim1[0] * scale)
        target_dim_1 = int(dim1[1] * scale)
        # numpy slicing clips at the image edge
        x1 = min(start_dim_0 + target_dim_0, dim1[0])
        y1 = min(start_dim_1 + target_dim_1, dim1[1])
        return self.ctx.fisheye_undistort(img, scaled_K, self.D, new_K,
                                          (start_dim_0, start_dim_1, x1 - start_dim_0, y1 - start_dim_1), to_gray=to_gray)

    def to_gray(self, img):
        """cv2.cvtColor(image, cv2.COLOR_BGR2GRAY) (position_finder.py:171)."""
        return self.ctx.cvt_gray(img)
