"""Configuration tree for the MonoFlex hot path.

Mirrors the keys of the reference's yacs tree that parameterise the path
(reference config/defaults.py:8-347, overridden by runs/monoflex.yaml:1-84) so that the
reference's experiment file drives this build unchanged.  yacs itself is not a
dependency: `CfgNode` below is a small attribute-dict with the same
merge_from_file / merge_from_list / freeze / clone surface.
"""
import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError("config is frozen; cannot set %s" % k)
        self[k] = v

    def _merge(self, other, path=""):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self:
                    self[k] = CfgNode()
                self[k]._merge(v, path + k + ".")
            else:
                if isinstance(v, str):
                    try:                      # yaml yields '("Car", "Pedestrian")' as a string
                        v = ast.literal_eval(v)
                    except Exception:
                        pass
                self[k] = v

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except Exception:
                    pass
            node[parts[-1]] = val

    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def clone(self):
        c = copy.deepcopy(self)
        c.defrost()
        return c

    def __deepcopy__(self, memo):
        c = CfgNode()
        for k, v in self.items():
            dict.__setitem__(c, k, copy.deepcopy(v, memo))
        return c


def _defaults():
    C = CfgNode()
    C.MODEL = CfgNode(dict(
        DEVICE="cuda", WEIGHT="", PRETRAIN=False, USE_SYNC_BN=False, REDUCE_LOSS_NORM=True,
        NORM="BN", INPLACE_ABN=False,
        # build-specific: arithmetic type of the HIP path ("fp32" parity mode | "bf16" perf mode)
        COMPUTE_DTYPE="fp32",
    ))
    C.MODEL.BACKBONE = CfgNode(dict(CONV_BODY="dla34", FREEZE_CONV_BODY_AT=0, DOWN_RATIO=4))
    C.MODEL.HEAD = CfgNode(dict(
        PREDICTOR="Base_Predictor", NUM_CHANNEL=256, USE_NORMALIZATION="BN", BN_MOMENTUM=0.1,
        REGRESSION_HEADS=[['2d_dim'], ['3d_offset'], ['3d_dim'], ['ori_cls', 'ori_offset'], ['depth']],
        REGRESSION_CHANNELS=[[4], [2], [3], [4, 2], [1]],
        LOSS_TYPE=["Penalty_Reduced_FocalLoss", "L1", "giou", "berhu"], HEATMAP_TYPE='centernet',
        LOSS_ALPHA=0.25, LOSS_GAMMA=2, LOSS_PENALTY_ALPHA=2, LOSS_BETA=4,
        MODIFY_INVALID_KEYPOINT_DEPTH=False, UNCERTAINTY_INIT=True, UNCERTAINTY_RANGE=[-10, 10],
        UNCERTAINTY_WEIGHT=1.0, KEYPOINT_LOSS='L1', KEYPOINT_NORM_FACTOR=1.0, CORNER_LOSS_DEPTH='direct',
        KEYPOINT_XY_WEIGHT=[1, 1], DEPTH_FROM_KEYPOINT=False, KEYPOINT_TO_DEPTH_RELU=True,
        DEPTH_MODE='exp', DEPTH_RANGE=[0.1, 100], DEPTH_REFERENCE=(26.494627, 16.05988),
        REGRESSION_OFFSET_STAT=[-0.5844396972302358, 9.075032501413093],
        USE_UNCERTAINTY=False,
        LOSS_NAMES=['hm_loss', 'center_loss', 'bbox_loss', 'depth_loss', 'offset_loss', 'orien_loss',
                    'dims_loss', 'corner_loss'],
        LOSS_UNCERTAINTY=[True, True, True, False, False, True, True, True], INIT_LOSS_WEIGHT=[],
        ENABLE_EDGE_FUSION=False, EDGE_FUSION_KERNEL_SIZE=3, EDGE_FUSION_NORM='BN', EDGE_FUSION_RELU=False,
        TRUNCATION_OFFSET_LOSS='L1', TRUNCATION_OUTPUT_FUSION='replace', TRUNCATION_CLS=False,
        OUTPUT_DEPTH='direct',
        DIMENSION_MEAN=((3.8840, 1.5261, 1.6286), (0.8423, 1.7607, 0.6602), (1.7635, 1.7372, 0.5968)),
        DIMENSION_STD=((0.4259, 0.1367, 0.1022), (0.2349, 0.1133, 0.1427), (0.1766, 0.0948, 0.1242)),
        DIMENSION_REG=['linear', True, False], DIMENSION_WEIGHT=[1, 1, 1],
        INIT_P=0.01, CENTER_SAMPLE='center', CENTER_MODE='max',
    ))
    C.INPUT = CfgNode(dict(
        HEIGHT_TRAIN=384, WIDTH_TRAIN=1280, HEIGHT_TEST=384, WIDTH_TEST=1280,
        PIXEL_MEAN=[0.485, 0.456, 0.406], PIXEL_STD=[0.229, 0.224, 0.225], TO_BGR=False,
        MODIFY_ALPHA=False, USE_APPROX_CENTER=False, HEATMAP_CENTER='3D', ADJUST_DIM_HEATMAP=False, ADJUST_BOUNDARY_HEATMAP=False,
        HEATMAP_RATIO=0.5, ELLIP_GAUSSIAN=False, IGNORE_DONT_CARE=False, ALLOW_OUTSIDE_CENTER=False,
        KEYPOINT_VISIBLE_MODIFY=False, APPROX_3D_CENTER='intersect',
        ORIENTATION='head-axis', ORIENTATION_BIN_SIZE=4, AUG_PARAMS=[[0.5]],
    ))
    C.DATASETS = CfgNode(dict(
        TRAIN=(), TEST=(), TRAIN_SPLIT="", TEST_SPLIT="", DETECT_CLASSES=("Car", "Pedestrian", "Cyclist"),
        FILTER_ANNO_ENABLE=False, FILTER_ANNOS=[0.9, 20], USE_RIGHT_IMAGE=False,
        CONSIDER_OUTSIDE_OBJS=False, MAX_OBJECTS=40, MIN_RADIUS=0.0, MAX_RADIUS=0.0,
        CENTER_RADIUS_RATIO=0.1,
    ))
    C.SOLVER = CfgNode(dict(
        OPTIMIZER="adamw", BASE_LR=3e-3, WEIGHT_DECAY=1e-5, MAX_ITERATION=30000, MAX_EPOCHS=70,
        DECAY_EPOCH_STEPS=[35, 45], LR_DECAY=0.1, LR_WARMUP=False, WARMUP_STEPS=-1, GRAD_NORM_CLIP=-1,
        BIAS_LR_FACTOR=2.0, BACKBONE_LR_FACTOR=1.0, IMS_PER_BATCH=32, EVAL_INTERVAL=2000,
        EVAL_AND_SAVE_EPOCH=False, EVAL_EPOCH_INTERVAL=2, SAVE_CHECKPOINT_EPOCH_INTERVAL=5,
        SAVE_CHECKPOINT_INTERVAL=1000, LOAD_OPTIMIZER_SCHEDULER=True,
        # keys the reference's entry script / scheduler read (config/defaults.py:256-299)
        MOMS=[0.95, 0.85], PCT_START=0.4, DIV_FACTOR=10, STEPS=(20000, 25000), LR_CLIP=0.0000001, WARMUP_EPOCH=1,
        GRAD_CLIP_FACTOR=99, GRAD_ALPHA=0.9, MASTER_BATCH=-1, MOMENTUM=0.9,
    ))
    C.DATALOADER = CfgNode(dict(NUM_WORKERS=8, SIZE_DIVISIBILITY=0, ASPECT_RATIO_GROUPING=False))
    C.TEST = CfgNode(dict(
        SINGLE_GPU_TEST=True, IMS_PER_BATCH=1, PRED_2D=True, UNCERTAINTY_AS_CONFIDENCE=False,
        METRIC=['R40'], EVAL_DIS_IOUS=False, EVAL_DEPTH=False, DETECTIONS_PER_IMG=50,
        DETECTIONS_THRESHOLD=0.1, VISUALIZE_THRESHOLD=0.4,
    ))
    C.OUTPUT_DIR = "./tools/logs"
    C.SEED = -1
    C.START_TIME = 0
    return C


cfg = _defaults()


def get_cfg(config_file=None, opts=None):
    """Fresh config = defaults <- yaml file <- KEY VALUE list (reference tools/plain_train_net.py:86-88)."""
    c = _defaults()
    if config_file:
        c.merge_from_file(config_file)
    if opts:
        c.merge_from_list(list(opts))
    return c


TYPE_ID_CONVERSION = {'Car': 0, 'Pedestrian': 1, 'Cyclist': 2, 'Van': -4, 'Truck': -4,
                      'Person_sitting': -2, 'Tram': -99, 'Misc': -99, 'DontCare': -1}
