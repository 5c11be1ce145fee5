"""Configuration object with the attribute names of the reference's utils/config.py.

The hot-path code reads ``config.<name>`` exactly like the reference's blocks / datasets do
(models/blocks.py:523-544, datasets/common.py:468,501,518), so a reference ``Config`` subclass
(e.g. train_DALES_PseudoLabel.py:44-201) can be passed in unchanged; this class exists so that
the package is usable without the reference tree.  Defaults follow utils/config.py:35-189; the
derived fields follow Config.__init__ (:191-233).  parameters.txt save/load (:235-445) is file
I/O outside the hot path and is not provided.
"""
import numpy as np


class Config:
    # ---- input
    dataset = ''
    dataset_task = ''
    num_classes = 0
    in_points_dim = 3
    in_features_dim = 1
    in_radius = 1.0
    input_threads = 8
    # ---- model
    architecture = []
    equivar_mode = ''
    invar_mode = ''
    first_features_dim = 64
    use_batch_norm = True
    batch_norm_momentum = 0.99
    segmentation_ratio = 1.0
    # ---- KPConv
    num_kernel_points = 15
    first_subsampling_dl = 0.02
    conv_radius = 2.5
    deform_radius = 5.0
    KP_extent = 1.0
    KP_influence = 'linear'
    aggregation_mode = 'sum'
    fixed_kernel_points = 'center'
    modulated = False
    n_frames = 1
    max_in_points = 0
    val_radius = 51.0
    max_val_points = 50000
    # ---- training
    learning_rate = 1e-3
    momentum = 0.9
    lr_decays = {200: 0.2, 300: 0.2}
    grad_clip_norm = 100.0
    augment_scale_anisotropic = True
    augment_scale_min = 0.9
    augment_scale_max = 1.1
    augment_symmetries = [False, False, False]
    augment_rotation = 'vertical'
    augment_noise = 0.005
    augment_color = 0.7
    augment_occlusion = 'none'
    augment_occlusion_ratio = 0.2
    augment_occlusion_num = 1
    weight_decay = 1e-3
    segloss_balance = 'none'
    class_w = []
    deform_fitting_mode = 'point2point'
    deform_fitting_power = 1.0
    deform_lr_factor = 0.1
    repulse_extent = 1.0
    batch_num = 10
    val_batch_num = 10
    max_epoch = 1000
    epoch_steps = 1000
    validation_size = 100
    checkpoint_gap = 50
    dropout = 0
    saving = True
    saving_path = None

    def __init__(self):
        arch = self.architecture
        self.num_layers = len([b for b in arch if 'pool' in b or 'strided' in b]) + 1
        # which layers contain a deformable convolution (utils/config.py:197-233)
        self.deform_layers = []
        layer_blocks = []
        for block in arch:
            if not any(tag in block for tag in ('pool', 'strided', 'global', 'upsample')):
                layer_blocks.append(block)
                continue
            deform = bool(layer_blocks) and bool(np.any(['deformable' in b for b in layer_blocks]))
            if ('pool' in block or 'strided' in block) and 'deformable' in block:
                deform = True
            self.deform_layers.append(deform)
            layer_blocks = []
            if 'global' in block or 'upsample' in block:
                break


_PL_ARCH = ['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb',
            'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb',
            'nearest_upsample', 'unary', 'nearest_upsample', 'unary',
            'nearest_upsample', 'unary', 'nearest_upsample', 'unary']


class DALESPLConfig(Config):
    """values of train_DALES_PseudoLabel.py:44-201 that reach the hot path"""
    dataset = 'DALESPL'
    contrast_start = 0        # train_DALES_PseudoLabel.py:181-182
    contrast_thd = 10
    input_threads = 10
    architecture = list(_PL_ARCH)
    num_kernel_points = 15
    in_radius = 18
    first_subsampling_dl = 0.4
    conv_radius = 2.5
    deform_radius = 5.0
    KP_extent = 1.0
    first_features_dim = 128
    in_features_dim = 3
    modulated = False
    use_batch_norm = True
    batch_norm_momentum = 0.02
    repulse_extent = 1.2
    learning_rate = 0.001
    momentum = 0.98
    grad_clip_norm = 100.0
    batch_num = 4
    dropout = 0.5
    class_w = [1, 1, 1, 1, 1, 1, 1, 1, 1]


class Vaihingen3DPLConfig(Config):
    """values of train_Vaihingen3D_PseudoLabel.py that reach the hot path"""
    dataset = 'Vaihingen3DPL'
    input_threads = 10
    architecture = list(_PL_ARCH)
    num_kernel_points = 15
    in_radius = 24
    first_subsampling_dl = 0.24
    conv_radius = 2.5
    deform_radius = 6.0
    KP_extent = 1.0
    first_features_dim = 64
    in_features_dim = 4
    modulated = False
    use_batch_norm = True
    batch_norm_momentum = 0.02
    repulse_extent = 1.2
    learning_rate = 0.01
    momentum = 0.98
    grad_clip_norm = 100.0
    batch_num = 4
    dropout = 0.5
    class_w = [1, 1, 1, 1, 1, 1, 1, 1, 1]


class DALESDeformConfig(DALESPLConfig):
    """BASELINE config 5 as SURVEY.md section 8d specifies it: the DALES network with every `resnetb` block replaced by
    `resnetb_deformable` (learned offsets, models/blocks.py:244-325) and `modulated = True` (:256,:366-367),
    deform_radius 5.0 -- so every level is searched with the deformable radius r * deform_radius / conv_radius = 2 r
    (datasets/common.py:498-503: about 8 x the neighbours) -- and feature rows / weights bf16 with fp32 accumulate
    (`feature_dtype`; the reference has no reduced-precision path: fp32 masters, bf16 rows in HBM)."""
    dataset = 'DALESDeform'
    architecture = [b.replace('resnetb', 'resnetb_deformable') for b in _PL_ARCH]
    modulated = True
    deform_radius = 5.0
    feature_dtype = 'bf16'


class DALESDeformF32Config(DALESDeformConfig):
    """the same network with f32 rows (A/B of the bf16 path; oracle comparisons at 1e-4)"""
    dataset = 'DALESDeformF32'
    feature_dtype = 'f32'


_WL_ARCH = ['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb',
            'nearest_upsample', 'nearest_upsample']


class Vaihingen3DWLConfig(Config):
    """values of train_Vaihingen3D_WeakLabel.py:46-178 that reach the weak-label step (BASELINE config 1): the 3-layer
    multi-path region-mining network KPFCNN_mprm, 64 first features, dl0 0.24, the region loss, gradient-NORM clipping at 1"""
    dataset = 'Vaihingen3DWL'
    input_threads = 10
    architecture = list(_WL_ARCH)
    num_kernel_points = 15
    in_radius = 18
    sub_radius = 5
    first_subsampling_dl = 0.24
    conv_radius = 2.5
    deform_radius = 1.0
    KP_extent = 1.0
    KP_influence = 'linear'
    aggregation_mode = 'sum'
    first_features_dim = 64
    in_features_dim = 4
    modulated = False
    use_batch_norm = True
    batch_norm_momentum = 0.02
    deform_fitting_mode = 'point2point'
    deform_fitting_power = 1.0
    deform_lr_factor = 0.1
    repulse_extent = 1.2
    max_epoch = 80
    learning_rate = 0.01
    momentum = 0.98
    lr_decays = {i: 0.98 for i in range(1, 1000)}
    grad_clip_norm = 1
    batch_num = 3
    class_w = [1, 1, 1, 1, 1, 1, 1, 1, 1]
    num_classes = 9
    model_name = 'KPFCNN_mprm'
    loss_type = 'region_mprm_loss'
    anchor_method = 'reduced'
