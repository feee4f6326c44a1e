"""Config handling: accepts the reference's YAML schema and derives the hot-path constants.

The reference reads one nested dict everywhere (train.py:502-503).  The keys
the inference path consumes are listed in SURVEY.md section 5; this module takes
the same dict (or a YAML file in the same schema, BOM tolerated) and derives
grid / feature-map / anchor counts once, the way the reference derives them:
  grid          = round((range_max - range_min) / voxel_size)   load_data.py:612-615, 2599-2601
  feature map   = grid[:2] // (layer_strides[0] // upsample_strides[0])   load_data.py:3019-3023
"""
import copy
import io

import numpy as np
import yaml

# The shipped reference configuration (configs/train.yaml:100-199), hot-path keys only.
_PEDESTRIAN_D435I = {
    "measure_time_extended": False,
    "eval_input_reader": {
        "batch_size": 1,
        "anchor_area_threshold": 1,
        "feature_map_size": [1, 64, 80],
        "num_point_features": 3,
        "desired_objects": ["Pedestrian"],
    },
    "model": {"second": {
        "voxel_generator": {
            "point_cloud_range": [0, -2.56, -3.0, 6.40, 2.56, 3.0],
            "voxel_size": [0.08, 0.08, 4.0],
            "max_number_of_points_per_voxel": 50,
            "max_number_of_voxels": 12000,
        },
        "num_class": 1,
        "voxel_feature_extractor": {"num_filters": 128, "with_distance": False},
        "rpn": {
            "layer_nums": [3, 5, 5],
            "layer_strides": [1, 2, 2],
            "num_filters": [64, 128, 256],
            "upsample_strides": [1, 2, 4],
            "num_upsample_filters": [128, 128, 128],
            "use_groupnorm": False,
            "num_groups": 32,
        },
        # training-side keys (configs/train.yaml:147-167, :143), read by Engine.head_loss only
        "encode_rad_error_by_sin": True,
        "loss": {
            "classification_loss": {"weighted_sigmoid_focal": {"alpha": 0.25, "gamma": 2.0, "anchorwise_output": True}},
            "localization_loss": {"weighted_smooth_l1": {"sigma": 3.0, "code_weight": [1.0] * 7}},
            "classification_weight": 1.0,
            "localization_weight": 1.5,
        },
        "pos_class_weight": 1.0,
        "neg_class_weight": 1.0,
        "loss_norm_type": "NormByNumPositives",
        "direction_loss_weight": 0.5,
        "use_sigmoid_score": True,
        "encode_background_as_zeros": True,
        "use_direction_classifier": True,
        "use_multi_class_nms": False,
        "nms_pre_max_size": 100,
        "nms_post_max_size": 50,
        "nms_score_threshold": 0.0,
        "nms_iou_threshold": 0.5,
        "num_point_features": 3,
        "target_assigner": {"anchor_generators": {"anchor_generator_stride": {
            "sizes": [0.6, 0.8, 1.73],
            "strides": [0.08, 0.08, 0.0],
            "offsets": [0.08, -2.56, -1.465],
            "rotations": [0, 1.57],
            "matched_threshold": 0.5,
            "unmatched_threshold": 0.35,
        }}},
    }},
}


def pedestrian_d435i_config(batch_size=1, max_points=None, max_voxels=None):
    """cfg-A: the reference's shipped d435i pedestrian config."""
    cfg = copy.deepcopy(_PEDESTRIAN_D435I)
    cfg["eval_input_reader"]["batch_size"] = batch_size
    vg = cfg["model"]["second"]["voxel_generator"]
    if max_points is not None:
        vg["max_number_of_points_per_voxel"] = max_points
    if max_voxels is not None:
        vg["max_number_of_voxels"] = max_voxels
    return cfg


def kitti_shaped_config(batch_size=1, num_class=1):
    """cfg-K: KITTI-shaped grid reachable by config only (BASELINE.json configs[2]):
    0.16 m pillars, 432x496 BEV, 4 point features, C=64, T=100, strides [2,2,2].
    num_class=2: "Pedestrian+Cyclist" as SURVEY section 8a sizes it (2 anchors per location, 22 head channels:
    both classes score every anchor, the label is the argmax)."""
    cfg = copy.deepcopy(_PEDESTRIAN_D435I)
    cfg["eval_input_reader"].update(batch_size=batch_size, num_point_features=4,
                                    feature_map_size=[1, 248, 216])
    s = cfg["model"]["second"]
    s["voxel_generator"] = {
        "point_cloud_range": [0, -39.68, -3.0, 69.12, 39.68, 1.0],
        "voxel_size": [0.16, 0.16, 4.0],
        "max_number_of_points_per_voxel": 100,
        "max_number_of_voxels": 12000,
    }
    s["num_point_features"] = 4
    s["num_class"] = int(num_class)
    if num_class > 1:
        cfg["eval_input_reader"]["desired_objects"] = ["Pedestrian", "Cyclist"][:num_class]
    s["voxel_feature_extractor"]["num_filters"] = 64
    s["rpn"].update(layer_strides=[2, 2, 2], upsample_strides=[1, 2, 4])
    s["target_assigner"]["anchor_generators"]["anchor_generator_stride"].update(
        strides=[0.32, 0.32, 0.0], offsets=[0.16, -39.52, -1.465])
    return cfg


def tiny_config(batch_size=1):
    """A 20x16 grid with narrow layers: small enough for loop-level CPU checks."""
    cfg = copy.deepcopy(_PEDESTRIAN_D435I)
    cfg["eval_input_reader"].update(batch_size=batch_size, feature_map_size=[1, 16, 20])
    s = cfg["model"]["second"]
    s["voxel_generator"].update(point_cloud_range=[0, -0.64, -3.0, 1.60, 0.64, 3.0],
                                max_number_of_points_per_voxel=8, max_number_of_voxels=500)
    s["voxel_feature_extractor"]["num_filters"] = 32
    s["rpn"].update(layer_nums=[1, 1, 1], num_filters=[32, 32, 64], num_upsample_filters=[32, 32, 32])
    s["target_assigner"]["anchor_generators"]["anchor_generator_stride"].update(
        offsets=[0.08, -0.64, -1.465])
    return cfg


def load_yaml(path):
    """Parses a reference-schema YAML file (train.py:502-503 uses yaml.FullLoader).
    The shipped configs/train.yaml starts with a UTF-8 BOM; 'utf-8-sig' strips it."""
    with io.open(path, "r", encoding="utf-8-sig") as f:
        return yaml.load(f.read(), Loader=yaml.FullLoader)


class Derived:
    """Constants derived once from the reference-schema config dict."""

    def __init__(self, config):
        self.config = config
        s = config["model"]["second"]
        vg = s["voxel_generator"]
        # float64 arrays, as the data loader builds them (load_data.py:2573-2574)
        self.pc_range = np.array(vg["point_cloud_range"], dtype=np.float64)
        self.voxel_size = np.array(vg["voxel_size"]).astype(np.float64)
        g = (self.pc_range[3:] - self.pc_range[:3]) / self.voxel_size
        self.grid = np.round(g).astype(np.int64)  # (nx, ny, nz)
        self.nx, self.ny, self.nz = (int(v) for v in self.grid)
        self.max_points = int(vg["max_number_of_points_per_voxel"])
        self.max_voxels = int(vg["max_number_of_voxels"])
        self.num_point_features = int(s["num_point_features"])
        # model/pointpillars.py:185-188: one more PFN input feature (the point's Euclidean norm)
        self.with_distance = bool(s["voxel_feature_extractor"].get("with_distance", False))
        self.pfn_in = self.num_point_features + 5 + (1 if self.with_distance else 0)
        self.pfn_filters = int(s["voxel_feature_extractor"]["num_filters"])
        r = s["rpn"]
        self.layer_nums = [int(v) for v in r["layer_nums"]]
        self.layer_strides = [int(v) for v in r["layer_strides"]]
        self.num_filters = [int(v) for v in r["num_filters"]]
        self.upsample_strides = [int(v) for v in r["upsample_strides"]]
        self.num_upsample_filters = [int(v) for v in r["num_upsample_filters"]]
        if len(self.layer_nums) != 3:
            raise ValueError("rpn.layer_nums must have 3 entries (model/voxelnet.py:557)")
        if bool(r.get("use_groupnorm", False)):
            raise NotImplementedError("use_groupnorm is unused in the reference and not built")
        factors = []
        for i in range(3):
            total = int(np.prod(self.layer_strides[:i + 1]))
            if total % self.upsample_strides[i] != 0:
                raise ValueError("layer stride product must be divisible by the upsample stride")
            factors.append(total // self.upsample_strides[i])
        if any(f != factors[0] for f in factors):
            raise ValueError("all blocks must upsample to the same map (model/voxelnet.py:568)")
        self.out_size_factor = self.layer_strides[0] // self.upsample_strides[0]
        self.head_h = self.ny // self.out_size_factor
        self.head_w = self.nx // self.out_size_factor
        self.feature_map_size = [1, self.head_h, self.head_w]
        ag = s["target_assigner"]["anchor_generators"]["anchor_generator_stride"]
        self.anchor_cfg = ag
        n_sizes = int(np.array(ag["sizes"]).reshape([-1, 3]).shape[0])
        self.num_anchor_per_loc = len(ag["rotations"]) * n_sizes
        self.num_class = int(s["num_class"])
        if not s["encode_background_as_zeros"] or s["use_multi_class_nms"]:
            raise NotImplementedError(
                "only encode_background_as_zeros / single-pass NMS is implemented "
                "in the reference's predict() (model/voxelnet.py:1152-1157, :1167 are TF stubs)")
        # model/voxelnet.py:690,714,1093,1297: without it the RPN has no conv_dir_cls and predict() does not flip
        self.use_direction_classifier = bool(s["use_direction_classifier"])
        # num_class > 1 is a TF stub in the reference's predict() (model/voxelnet.py:1183-1185: reduce_max /
        # argmax over the class scores of an anchor); built here as that documented extension.  The fused head
        # row holds 32 columns.
        cols = self.num_anchor_per_loc * (7 + self.num_class + (2 if self.use_direction_classifier else 0))
        if self.num_class < 1 or cols > 32:
            raise NotImplementedError(f"num_anchor_per_loc * (7 + num_class + 2) = {cols} head columns > 32")
        self.num_anchors = self.head_h * self.head_w * self.num_anchor_per_loc
        self.nms_pre_max_size = int(s["nms_pre_max_size"])
        self.nms_post_max_size = int(s["nms_post_max_size"])
        self.nms_score_threshold = float(s["nms_score_threshold"])
        self.nms_iou_threshold = float(s["nms_iou_threshold"])
        er = config["eval_input_reader"]
        self.batch_size = int(er["batch_size"])
        self.anchor_area_threshold = er["anchor_area_threshold"]
        self.concat_channels = int(sum(self.num_upsample_filters))

    def rpn_dict(self):
        return {"layer_nums": self.layer_nums, "layer_strides": self.layer_strides,
                "num_filters": self.num_filters, "upsample_strides": self.upsample_strides,
                "num_upsample_filters": self.num_upsample_filters}

    def model_dict(self):
        return {"voxel_size": self.voxel_size, "pc_range": self.pc_range, "grid": self.grid,
                "rpn": self.rpn_dict(), "with_distance": self.with_distance}

    def nms_dict(self):
        return {"nms_score_threshold": self.nms_score_threshold, "nms_pre_max_size": self.nms_pre_max_size,
                "nms_post_max_size": self.nms_post_max_size, "nms_iou_threshold": self.nms_iou_threshold,
                "num_class": self.num_class, "use_direction_classifier": self.use_direction_classifier}
