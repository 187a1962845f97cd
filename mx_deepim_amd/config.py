"""Constants of the hot path, mirroring the reference's config defaults + the shipped YAML
(deepim/config/config.py:11-118, experiments/deepim/cfgs/deepim_flownet_LM_SIXD_v1_ape_RFMx4_8epoch.yaml).
Only the keys the render-and-compare inner loop reads are kept."""
import copy

import numpy as np


class AttrDict(dict):
    """easydict-style attribute access (easydict itself is not a dependency here)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def default_config():
    cfg = AttrDict()
    cfg.dataset = AttrDict(
        INTRINSIC_MATRIX=np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]], dtype=np.float32),
        NORMALIZE_FLOW=20.0,
        NORMALIZE_3D_POINT=0.1,
        trans_means=np.array([0.0, 0.0, 0.0]),
        trans_stds=np.array([1.0, 1.0, 1.0]),
    )
    cfg.network = AttrDict(
        PIXEL_MEANS=np.array([123.68, 116.779, 103.939], dtype=np.float32),
        INPUT_MASK=True,
        INPUT_DEPTH=False,
        PRED_FLOW=True,
        PRED_MASK=True,
        ROT_TYPE="QUAT",
        ROT_COORD="CAMERA",
        REGRESSOR_NUM=1,
        STANDARD_FLOW_REP=False,
        TRAIN_ITER=True, TRAIN_ITER_SIZE=4,   # yaml :57-58 — refinement iterations inside one training step (module.py:1131-1137)
        X3_CONV=False,     # split-fp16 conv path: fp32-grade accuracy (≈1e-6) on the fp16 matrix cores (not a reference key)
        FP16_CONV=False,   # BASELINE config 5: fp16 conv path (not a reference key; the reference is fp32 only)
    )
    cfg.train_iter = AttrDict(SE3_PM_LOSS=True, SE3_PM_LOSS_TYPE="L1", LW_PM=0.1, LW_FLOW=0.25, LW_MASK=0.03,
                              NUM_3D_SAMPLE=3000, SE3_PM_SL1_SCALAR=1.0, SE3_DIST_LOSS=False,
                              LW_ROT=0.0, LW_TRANS=0.0, TRANS_LOSS_TYPE="L2", TRANS_SMOOTH_L1_SCALAR=3.0)   # config.py:104-108
    # experiments/deepim/cfgs/deepim_flownet_LM_SIXD_v1_ape_RFMx4_8epoch.yaml:76-92 (keys the label generation reads)
    # (the yaml also sets MASK_DILATE: True — a random cv2 dilation, i.e. loader-side augmentation, not built here)
    cfg.TRAIN = AttrDict(INIT_MASK="box_gt", FLOW_WEIGHT_TYPE="viz", MASK_DILATE=False,
                         optimizer="sgd", lr=0.0001, momentum=0.975, wd=0.0005)   # config.py:68-77
    cfg.TEST = AttrDict(test_iter=4, FAST_TEST=True, UPDATE_MASK="box_rendered", INIT_MASK="box_rendered")
    cfg.SCALES = [(480, 640)]
    return cfg


ROT_COORD_CODE = {"model": 0, "camera": 1, "camera_new": 2, "naive": 3}
