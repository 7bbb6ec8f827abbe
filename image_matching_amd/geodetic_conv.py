"""WGS-84 geodetic <-> ECEF <-> local NED, the class the reference's facade uses for pixel <-> lat/lon
(reference geodetic_conv.py:6-126; callers match_finder.py:59-61,272-274). Host-side double math;
the same public method names, argument order and conventions (degrees in and out, `down` is negated
on the way into ECEF exactly as the reference does, geodetic2Ned returns [north, east, down])."""
import math

import numpy as np

_A = 6378137.0
_B = 6356752.3142
_E1SQ = 6.69437999014e-3
_E2SQ = 6.73949674228e-3


def _rot_ned(lat, lon):
    sl, cl, so, co = math.sin(lat), math.cos(lat), math.sin(lon), math.cos(lon)
    return np.array([[-sl * co, -sl * so, cl], [-so, co, 0.0], [cl * co, cl * so, sl]])


class GeodeticConvert:
    def __init__(self):
        self.have_reference_ = False
        self.kSemimajorAxis = _A
        self.kSemiminorAxis = _B
        self.kFirstEccentricitySquared = _E1SQ
        self.kSecondEccentricitySquared = _E2SQ
        self.kFlattening = 1 / 298.257223563

    def isInitialised(self):
        return self.have_reference_

    def getReference(self):
        if self.have_reference_:
            return [self.initial_latitude_, self.initial_longitude_, self.initial_altitude_]

    def deg2Rad(self, d):
        return d / 180.0 * math.pi

    def rad2Deg(self, r):
        return (r / math.pi) * 180

    def initialiseReference(self, latitude, longitude, altitude):
        self.initial_latitude_ = self.deg2Rad(latitude)
        self.initial_longitude_ = self.deg2Rad(longitude)
        self.initial_altitude_ = altitude
        x, y, z = self.geodetic2Ecef(latitude, longitude, altitude)
        self.initial_ecef_x_, self.initial_ecef_y_, self.initial_ecef_z_ = x, y, z
        geocentric_lat = math.atan2(z, math.hypot(x, y))  # ECEF->NED uses the geocentric latitude
        self.ecef_to_ned_matrix_ = _rot_ned(geocentric_lat, self.initial_longitude_)
        self.ned_to_ecef_matrix_ = _rot_ned(self.initial_latitude_, self.initial_longitude_).T
        self.have_reference_ = True

    def geodetic2Ecef(self, latitude, longitude, altitude):
        la, lo = self.deg2Rad(latitude), self.deg2Rad(longitude)
        xi = math.sqrt(1 - _E1SQ * math.sin(la) * math.sin(la))
        x = (_A / xi + altitude) * math.cos(la) * math.cos(lo)
        y = (_A / xi + altitude) * math.cos(la) * math.sin(lo)
        z = (_A / xi * (1 - _E1SQ) + altitude) * math.sin(la)
        return [x, y, z]

    def ecef2Geodetic(self, x, y, z):
        # closed form (Zhu / Heikkinen), the same formula the reference evaluates
        r = math.sqrt(x * x + y * y)
        esq = _A * _A - _B * _B
        f = 54 * _B * _B * z * z
        g = r * r + (1 - _E1SQ) * z * z - _E1SQ * esq
        c = (_E1SQ * _E1SQ * f * r * r) / (g ** 3)
        s = (1 + c + math.sqrt(c * c + 2 * c)) ** (1.0 / 3.0)
        p = f / (3 * ((s + 1 / s + 1) ** 2) * g * g)
        q = math.sqrt(1 + 2 * _E1SQ * _E1SQ * p)
        r0 = -(p * _E1SQ * r) / (1 + q) + math.sqrt(
            0.5 * _A * _A * (1 + 1.0 / q) - p * (1 - _E1SQ) * z * z / (q * (1 + q)) - 0.5 * p * r * r)
        u = math.sqrt((r - _E1SQ * r0) ** 2 + z * z)
        v = math.sqrt((r - _E1SQ * r0) ** 2 + (1 - _E1SQ) * z * z)
        z0 = _B * _B * z / (_A * v)
        altitude = u * (1 - _B * _B / (_A * v))
        latitude = self.rad2Deg(math.atan((z + _E2SQ * z0) / r))
        longitude = self.rad2Deg(math.atan2(y, x))
        return latitude, longitude, altitude

    def ned2Ecef(self, north, east, down):
        ret = self.ned_to_ecef_matrix_.dot(np.array([north, east, -down], dtype=np.float64))
        return ret[0] + self.initial_ecef_x_, ret[1] + self.initial_ecef_y_, ret[2] + self.initial_ecef_z_

    def ned2Geodetic(self, north, east, down):
        return self.ecef2Geodetic(*self.ned2Ecef(north, east, down))

    def ecef2Ned(self, x, y, z):
        v = np.array([x - self.initial_ecef_x_, y - self.initial_ecef_y_, z - self.initial_ecef_z_], dtype=np.float64)
        ret = self.ecef_to_ned_matrix_.dot(v)
        return [ret[0], ret[1], ret[2]]

    def geodetic2Ned(self, latitude, longitude, altitude):
        return self.ecef2Ned(*self.geodetic2Ecef(latitude, longitude, altitude))
