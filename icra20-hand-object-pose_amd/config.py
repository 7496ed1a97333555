"""YAML configuration surface (config_autodataset.yaml of the reference, ConfigParser.cpp:30-137)."""
import os

import yaml

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "config_autodataset.yaml")

# keys the hot path reads (Hand.cpp:68-83,154-156,590-597; PoseEstimator.cpp:43,66-73,135-137,244-245,471-472)
REQUIRED = ["model_name", "object_symmetry", "hand_match", "lcp", "pose_estimator_high_confidence_thres", "icp_dist_thres",
            "icp_angle_thres", "super4pcs_sample_size", "super4pcs_overlap", "super4pcs_delta", "super4pcs_dispersion",
            "super4pcs_success_quadrilaterals", "super4pcs_max_normal_difference", "super4pcs_max_color_distance",
            "super4pcs_max_time_seconds"]


def load_config(path=None):
    with open(path or DEFAULT_CONFIG) as f:
        cfg = yaml.safe_load(f)
    missing = [k for k in REQUIRED if k not in cfg]
    if missing:
        raise KeyError(f"config is missing {missing}")
    if cfg["model_name"] not in cfg["object_symmetry"]:
        raise KeyError(f"object_symmetry has no entry for model_name={cfg['model_name']}")
    return cfg
