"""Parameter source for the facade. The reference reads every parameter with rospy.get_param
(match_finder.py:29-38) from a YAML that is not in its repository (param_publisher.py:7-20). Here:
rospy when it is importable and a master is reachable, otherwise this process-wide table.

Defaults: OpenCV's SIFT_create defaults, BASELINE.json's ratio 0.7 and frame width 640; the three
homography-gate values are assumptions (the flight YAML is missing) and are meant to be overridden."""

DEFAULTS = {
    "image_size_sift": 640,
    "points_quality_sift": 0.7,
    "nOctaveLayers_sift": 3,
    "contrastThreshold_sift": 0.04,
    "edgeThreshold_sift": 10,
    "sigma_sift": 1.6,
    "angle_restriction_homography": 0.35,
    "low_scale_restriction_homography": 0.5,
    "high_scale_restriction_homography": 2.0,
    "camera_pitch_angle": 0.0,
    # not a parameter of the reference: "sift" (the detector north_star names, the commented call of
    # match_finder.py:205-209) or "akaze" (the call the checked-in file makes, match_finder.py:210-211)
    "detector": "sift",
}

_table = dict(DEFAULTS)


def set_params(**kw):
    _table.update(kw)


def load_yaml(path):
    import yaml
    with open(path) as f:
        _table.update(yaml.safe_load(f) or {})


def get_param(name):
    try:
        import rospy
        return rospy.get_param(name)
    except ImportError:
        pass
    except Exception:
        if name not in _table:
            raise
    return _table[name]


def get_param_or_default(name):
    """For keys the reference does not have (a flight YAML will not hold them)."""
    try:
        return get_param(name)
    except Exception:
        return DEFAULTS[name]
