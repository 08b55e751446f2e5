"""Minimal configuration objects carrying exactly the keys the hot-path modules
read (values = configs/config_files/ISCNet_test.yaml; dataset constants =
configs/scannet_config.py:11-25).  The reference's CONFIG class (logging, save
dirs, CUDA_VISIBLE_DEVICES) is out of scope."""
import copy
import os

import numpy as np

DEFAULT_CONFIG = {
    'method': 'ISCNet',
    'seed': 10,
    'mode': 'demo',
    'data': {
        'num_point': 80000, 'num_target': 256, 'vote_factor': 1,
        'cluster_sampling': 'seed_fps', 'no_height': False,
        'use_color_detection': False, 'use_color_completion': False,
        'hidden_dim': 512, 'c_dim': 512, 'z_dim': 32, 'threshold': 0.5,
        'use_cls_for_completion': False, 'skip_propagate': True,
    },
    'model': {
        'backbone': {'method': 'Pointnet2Backbone', 'loss': 'Null'},
        'voting': {'method': 'VotingModule', 'loss': 'Null'},
        'detection': {'method': 'ProposalModule', 'loss': 'DetectionLoss'},
        'skip_propagation': {'method': 'SkipPropagation', 'loss': 'Null'},
        'completion': {'method': 'ONet', 'loss': 'ONet_Loss', 'weight': 0.005},
    },
    'test': {'phase': 'completion', 'batch_size': 1},
    'demo': {'phase': 'completion'},
    'generation': {
        'generate_mesh': True, 'resolution_0': 32, 'upsampling_steps': 0,
        'use_sampling': False, 'refinement_step': 0, 'simplify_nfaces': None,
        'dump_threshold': 0.5, 'dump_results': True,
    },
}


MEAN_SIZE_NPZ = os.path.join('datasets', 'scannet', 'scannet_means.npz')


class ScannetConfig(object):
    """configs/scannet_config.py:11-25: 8 classes, 12 heading bins, 8 size
    clusters.  The reference reads mean_size_arr from the DATA file
    `datasets/scannet/scannet_means.npz` relative to the working directory
    (scannet_config.py:21).  Same here: an explicit array or path wins
    (`mean_size_arr=`), then $RFD_MEAN_SIZE_NPZ, then that relative path; if none
    exists a neutral placeholder is used and `placeholder_sizes` is set --
    parse_predictions (box decoding, empty-box removal, 3-D NMS: everything behind
    `selection='nms'`) then RAISES like the reference's missing-file error, unless the
    caller opts in (eval config 'allow_placeholder_sizes'); paths that never decode boxes
    (`selection='all'`, the benchmark) are unaffected."""

    def __init__(self, mean_size_arr=None):
        self.num_class = 8
        self.num_heading_bin = 12
        self.num_size_cluster = 8
        self.placeholder_sizes = False
        if mean_size_arr is None:
            for cand in (os.environ.get('RFD_MEAN_SIZE_NPZ'), MEAN_SIZE_NPZ):
                if cand and os.path.exists(cand):
                    mean_size_arr = cand
                    break
        if isinstance(mean_size_arr, (str, bytes, os.PathLike)):
            mean_size_arr = np.load(mean_size_arr)['arr_0']
        if mean_size_arr is None:
            self.placeholder_sizes = True
            mean_size_arr = np.full((8, 3), 0.8)
        self.mean_size_arr = np.asarray(mean_size_arr, dtype=np.float64)
        assert self.mean_size_arr.shape == (self.num_size_cluster, 3), self.mean_size_arr.shape

    def class2angle_cuda(self, pred_cls, residual, to_label_format=True):
        """scannet_config.py:55-63"""
        angle_per_class = 2 * np.pi / float(self.num_heading_bin)
        angle = pred_cls.float() * angle_per_class + residual
        if to_label_format:
            angle = angle - 2 * np.pi * (angle > np.pi).float()
        return angle


class Config(object):
    def __init__(self, overrides=None, mean_size_arr=None):
        self.config = copy.deepcopy(DEFAULT_CONFIG)
        for sect, kv in (overrides or {}).items():
            if isinstance(kv, dict):
                self.config.setdefault(sect, {}).update(kv)
            else:
                self.config[sect] = kv
        self.dataset_config = ScannetConfig(mean_size_arr)
        self.eval_config = {'dataset_config': self.dataset_config}
        self._mount_eval_config()

    def _mount_eval_config(self):
        """configs/config_utils.py:131-149 mount_external_config: the `val` / `test` block of the YAML
        becomes the dictionary parse_predictions reads (`eval_overrides` = what differs from the defaults)."""
        ev = self.config.get('val', self.config.get('test')) or {}
        m = {}
        if 'faster_eval' in ev:
            m['remove_empty_box'] = not ev['faster_eval']
        for src, dst in (('use_3d_nms', 'use_3d_nms'), ('nms_iou', 'nms_iou'), ('use_old_type_nms', 'use_old_type_nms'),
                         ('use_cls_nms', 'cls_nms'), ('per_class_proposal', 'per_class_proposal'),
                         ('conf_thresh', 'conf_thresh')):
            if src in ev:
                m[dst] = ev[src]
        self.eval_overrides = m
        self.eval_config.update(m)

    @classmethod
    def from_yaml(cls, path, mode='demo', overrides=None, mean_size_arr=None):
        """Read one of the reference's own config files (configs/config_files/ISCNet_test.yaml: same keys,
        configs/config_utils.py:83-97 read_to_dict) and set `mode` as main.py:21 / config_utils.py:99-115 do.
        Keys this path does not use (device, log, dataset paths) are kept in `.config` untouched."""
        import yaml
        with open(path) as f:
            doc = yaml.safe_load(f) or {}
        cfg = cls(doc, mean_size_arr=mean_size_arr)
        cfg.config['mode'] = mode
        for sect, kv in (overrides or {}).items():
            if isinstance(kv, dict):
                cfg.config.setdefault(sect, {}).update(kv)
            else:
                cfg.config[sect] = kv
        if mode not in cfg.config or 'phase' not in (cfg.config.get(mode) or {}):
            raise KeyError("config %s has no `%s: {phase: ...}` block" % (path, mode))
        cfg._mount_eval_config()
        return cfg

    def log_string(self, s):
        print(s)
