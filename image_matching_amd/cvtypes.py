"""Stand-ins for the cv2 value types the reference's callers touch:
cv2.KeyPoint (.pt used at match_finder.py:231-232), cv2.DMatch (.distance match_finder.py:193,
.queryIdx/.trainIdx match_finder.py:231-232), and lazy containers so that a frame's keypoints and
descriptors stay on the device until somebody actually reads them."""
import numpy as np


class KeyPoint:
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, x=0.0, y=0.0, size=0.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
        self.pt = (float(x), float(y))
        self.size = float(size)
        self.angle = float(angle)
        self.response = float(response)
        self.octave = int(octave)
        self.class_id = int(class_id)

    def __repr__(self):
        return "KeyPoint(pt=(%.3f, %.3f), size=%.3f, angle=%.2f, response=%.5f, octave=%d)" % (
            self.pt[0], self.pt[1], self.size, self.angle, self.response, self.octave)


class DMatch:
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx=-1, trainIdx=-1, imgIdx=0, distance=float("inf")):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.imgIdx = int(imgIdx)
        self.distance = float(distance)

    def __repr__(self):
        return "DMatch(queryIdx=%d, trainIdx=%d, distance=%.3f)" % (self.queryIdx, self.trainIdx, self.distance)


class KeyPoints:
    """Immutable sequence of KeyPoint backed by one float32 [K,7] array
    (x, y, size, angle, response, octave, class_id). len() and indexing behave like the tuple
    cv2 returns (position_finder.py:257,394-395 only call len()); `.xy` gives all points at once."""

    def __init__(self, records):
        self.records = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 7)

    def __len__(self):
        return self.records.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return KeyPoints(self.records[i])
        r = self.records[i]
        return KeyPoint(r[0], r[1], r[2], r[3], r[4], int(r[5]), int(r[6]))

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    @property
    def xy(self):
        return self.records[:, :2]


class DeviceDescriptors:
    """Descriptor matrix that lives in HBM. Behaves like the float32 [K,128] ndarray cv2 returns when
    read (np.asarray, indexing, .shape, len) -- the rows are downloaded on first use -- and carries
    the device handle `features` that find_matches uses directly, so the reference's per-call
    re-upload (match_finder.py:187-188) disappears."""

    def __init__(self, features):
        self.features = features
        self._host = None

    @property
    def shape(self):
        return (len(self.features), 128)

    @property
    def dtype(self):
        return np.dtype(np.float32)

    @property
    def ndim(self):
        return 2

    def __len__(self):
        return len(self.features)

    def __array__(self, dtype=None, copy=None):
        if self._host is None:
            self._host = self.features.download(keypoints=False)[1]
        return self._host if dtype is None else self._host.astype(dtype, copy=False)

    def __getitem__(self, i):
        return self.__array__()[i]



class GoodMatches:
    """The `good` list find_matches returns -- [[DMatch], [DMatch], ...] (match_finder.py:191-194) -- backed by three
    arrays. len(), indexing and iteration behave like the list of one-element lists the reference builds (its callers
    use len(good) and m[0].queryIdx / m[0].trainIdx: position_finder.py:316, match_finder.py:231-232); the DMatch
    objects are made when somebody reads them, and find_keypoints_transform reads the arrays directly."""

    def __init__(self, query_idx, train_idx, distance):
        self.query_idx = np.ascontiguousarray(query_idx, dtype=np.int64)
        self.train_idx = np.ascontiguousarray(train_idx, dtype=np.int64)
        self.distance = np.ascontiguousarray(distance, dtype=np.float32)

    def __len__(self):
        return self.query_idx.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return GoodMatches(self.query_idx[i], self.train_idx[i], self.distance[i])
        return [DMatch(int(self.query_idx[i]), int(self.train_idx[i]), 0, float(self.distance[i]))]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return "GoodMatches(%d)" % len(self)


class FloatDescriptors(np.ndarray):
    """The float32 [K, d] descriptor matrix of the AKAZE path: an ndarray (what cv2 returns) that can carry the
    handle of its copy in HBM, made the first time the matrix serves as the train side of find_matches; the ROI's
    descriptors are matched against for up to 11 frames (position_finder.py:232-240) and travel once."""

    def __new__(cls, a, source=None):
        obj = np.ascontiguousarray(a, dtype=np.float32).view(cls)
        obj.fset = None
        # (context, serial) of the AKAZE extraction that produced this matrix: while it is the context's last extraction
        # its descriptors are still on the device and the query side of find_matches takes them there
        obj.source = source
        return obj

    def __array_finalize__(self, obj):
        self.fset = None      # slices and copies are new matrices
        self.source = None
