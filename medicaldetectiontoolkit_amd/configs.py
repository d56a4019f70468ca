"""Plain-attribute config object carrying the names the hot-path call sites read
(reference: default_configs.py:21, experiments/lidc_exp/configs.py:23-334).  Values default to
the LIDC experiment; BASELINE.json overrides patch_size_3D to 128^3."""
import numpy as np


class Configs(object):
    def __init__(self, dim=3, model="mrcnn", patch_size=None, batch_size=None, **overrides):
        self.dim = dim
        self.model = model
        # experiments/lidc_exp/configs.py:78-79 (BASELINE: 128^3), :115
        self.patch_size_2D = [288, 288]
        self.patch_size_3D = [128, 128, 128]
        if patch_size is not None:
            if dim == 2:
                self.patch_size_2D = list(patch_size)
            else:
                self.patch_size_3D = list(patch_size)
        self.patch_size = self.patch_size_2D if dim == 2 else self.patch_size_3D
        self.batch_size = batch_size if batch_size is not None else (20 if dim == 2 else 8)
        self.n_channels = 1
        self.n_workers = 8
        self.seed = 0
        # backbone (:100-107, default_configs.py:64)
        self.start_filts = 48 if dim == 2 else 18
        self.end_filts = self.start_filts * 4 if dim == 2 else self.start_filts * 2
        self.res_architecture = "resnet50"
        self.norm = None
        self.relu = "relu"
        self.weight_init = None
        self.sixth_pooling = False
        self.n_latent_dims = 0
        self.operate_stride1 = False
        # optimiser / schedule (:104, 113-114, 219)
        self.num_epochs = 100
        self.num_train_batches = 200 if dim == 2 else 200
        self.weight_decay = 0
        self.learning_rate = [1e-4] * self.num_epochs
        # classes (:137, 231, 234)
        self.class_dict = {1: "benign", 2: "malignant"}
        self.head_classes = 3
        self.num_seg_classes = 2
        self.class_specific_seg_flag = False
        self.frcnn_mode = False
        # test / ensembling (:131-147, default_configs.py:88)
        self.test_aug = True
        self.test_n_epochs = 5
        self.save_n_models = 5
        self.min_det_thresh = 0.1
        self.wcs_iou = 1e-5
        self.merge_2D_to_3D_preds = False
        self.return_masks_in_val = True
        self.return_masks_in_test = False
        self.n_plot_rpn_props = 5 if dim == 2 else 30
        # anchors / RPN (:237-260)
        self.backbone_strides = {"xy": [4, 8, 16, 32], "z": [1, 2, 4, 8]}
        self.rpn_anchor_scales = {"xy": [[8], [16], [32], [64]], "z": [[2], [4], [8], [16]]}
        self.pyramid_levels = [0, 1, 2, 3]
        self.n_rpn_features = 512 if dim == 2 else 128
        self.rpn_anchor_ratios = [0.5, 1, 2]
        self.rpn_anchor_stride = 1
        self.rpn_nms_threshold = 0.7
        self.rpn_train_anchors_per_image = 6
        self.train_rois_per_image = 6
        self.roi_positive_ratio = 0.5
        self.anchor_matching_iou = 0.7
        self.shem_poolsize = 10
        # heads (:266-293)
        self.pool_size = (7, 7) if dim == 2 else (7, 7, 3)
        self.mask_pool_size = (14, 14) if dim == 2 else (14, 14, 5)
        self.mask_shape = (28, 28) if dim == 2 else (28, 28, 10)
        self.rpn_bbox_std_dev = np.array([0.1, 0.1, 0.1, 0.2, 0.2, 0.2])
        self.bbox_std_dev = np.array([0.1, 0.1, 0.1, 0.2, 0.2, 0.2])
        self.pre_nms_limit = 3000 if dim == 2 else 6000
        self.roi_chunk_size = 2500 if dim == 2 else 600
        self.post_nms_rois_training = 500 if dim == 2 else 75
        self.post_nms_rois_inference = 500
        self.model_max_instances_per_batch_element = 10 if dim == 2 else 30
        self.detection_nms_threshold = 1e-5
        self.model_min_confidence = 0.1
        for k, v in overrides.items():
            setattr(self, k, v)
        self.finalize()

    def finalize(self):
        """Derived fields (:272-306, 313-332)."""
        ps = self.patch_size
        z = self.patch_size_3D[2]
        self.window = np.array([0, 0, ps[0], ps[1], 0, z])
        self.scale = np.array([ps[0], ps[1], ps[0], ps[1], z, z])
        if self.dim == 2:
            self.rpn_bbox_std_dev = self.rpn_bbox_std_dev[:4]
            self.bbox_std_dev = self.bbox_std_dev[:4]
            self.window = self.window[:4]
            self.scale = self.scale[:4]
            self.backbone_shapes = np.array(
                [[int(np.ceil(ps[0] / s)), int(np.ceil(ps[1] / s))] for s in self.backbone_strides["xy"]])
        else:
            self.backbone_shapes = np.array(
                [[int(np.ceil(ps[0] / s)), int(np.ceil(ps[1] / s)), int(np.ceil(ps[2] / sz))]
                 for s, sz in zip(self.backbone_strides["xy"], self.backbone_strides["z"])])
        if self.model in ("retina_net", "retina_unet"):
            if len(self.rpn_anchor_scales["xy"][0]) == 1:
                self.rpn_anchor_scales = {
                    "xy": [[ii[0], ii[0] * (2 ** (1 / 3)), ii[0] * (2 ** (2 / 3))] for ii in self.rpn_anchor_scales["xy"]],
                    "z": [[ii[0], ii[0] * (2 ** (1 / 3)), ii[0] * (2 ** (2 / 3))] for ii in self.rpn_anchor_scales["z"]]}
            self.n_anchors_per_pos = len(self.rpn_anchor_ratios) * 3
            self.n_rpn_features = 256 if self.dim == 2 else 64
            self.pre_nms_limit = 10000 if self.dim == 2 else 50000
            self.anchor_matching_iou = 0.5
            self.num_seg_classes = 3 if self.class_specific_seg_flag else 2
            if self.model == "retina_unet":
                self.operate_stride1 = True
        return self
