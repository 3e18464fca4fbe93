"""Occupancy mIoU metrics -- drop-ins for mmdet3d/datasets/occ_metrics.py:52-185 (Metric_mIoU) and
:413-594 (Metric_mIoU_Temporal): same constructor flags, add_batch / count_miou semantics.  The
18x18 confusion matrix is accumulated on the GPU (pw_confusion_hist, exact integer work); the
final per-class IoU / nanmean is the reference's numpy arithmetic."""
import numpy as np
import torch

from . import ops

CLASS_NAMES = ['others', 'barrier', 'bicycle', 'bus', 'car', 'construction_vehicle', 'motorcycle',
               'pedestrian', 'traffic_cone', 'trailer', 'truck', 'driveable_surface', 'other_flat',
               'sidewalk', 'terrain', 'manmade', 'vegetation', 'free']


def _dev(a, device):
    t = torch.as_tensor(a)
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    return t.to(device)


class Metric_mIoU:
    def __init__(self, save_dir='.', num_classes=18, use_lidar_mask=False, use_image_mask=False,
                 device='cuda:0'):
        self.class_names = CLASS_NAMES
        self.num_classes = num_classes
        self.use_lidar_mask, self.use_image_mask = use_lidar_mask, use_image_mask
        self.device = device
        self._hist = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=device)
        self._occ_hist = torch.zeros(2, 2, dtype=torch.int64, device=device)
        self.cnt = 0

    @property
    def hist(self):
        return self._hist.cpu().numpy().astype(np.float64)

    @property
    def occ_hist(self):
        return self._occ_hist.cpu().numpy().astype(np.float64)

    def add_batch(self, semantics_pred, semantics_gt, mask_lidar, mask_camera):
        self.cnt += 1
        mask = mask_camera if self.use_image_mask else (mask_lidar if self.use_lidar_mask else None)
        p = _dev(semantics_pred, self.device).to(torch.uint8)
        g = _dev(semantics_gt, self.device).to(torch.uint8)
        m = _dev(mask, self.device) if mask is not None else None
        ops.confusion_hist(p, g, m, self.num_classes, self._hist)
        # binary occupied/free histogram (occ_metrics.py:137-141)
        free = self.num_classes - 1
        ops.confusion_hist((p != free).to(torch.uint8), (g != free).to(torch.uint8), m, 2, self._occ_hist)

    @staticmethod
    def per_class_iu(hist):
        with np.errstate(divide='ignore', invalid='ignore'):
            return np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))

    def count_miou(self, verbose=False):
        mIoU = self.per_class_iu(self.hist)
        res = round(np.nanmean(mIoU[:self.num_classes - 1]) * 100, 2)
        if verbose:
            for i in range(self.num_classes):
                print('===> %s - IoU = %s' % (self.class_names[i], round(mIoU[i] * 100, 2)))
            print('===> mIoU of %d samples: %s' % (self.cnt, res))
        return self.class_names, mIoU, self.cnt, res

    def count_iou(self):
        IoU = self.per_class_iu(self.occ_hist)
        return ['free', 'occupied'], IoU, self.cnt, round(IoU[-1] * 100, 2)


class Metric_mIoU_Temporal:
    """Drop-in for occ_metrics.py:413-594, same call signatures and return values:

    * add_batch(semantics_pred, semantics_gt_temp, mask_lidar_temp, mask_camera_temp): `semantics_pred` is the stack of
      states {0,2,4,6} (apis/test.py:218-223); the three dicts are keyed by the ground-truth index idx in {0,2,4,6}
      (keyframes at 2 Hz = 0/1/2/3 s) and idx is scored against semantics_pred[idx // 2] (:505-510);
    * count_miou() -> (per-class IoU at 1 s, [mIoU 1 s, 2 s, 3 s]) (:548-575); count_iou() -> [IoU 1 s, 2 s, 3 s] (:577-594);
    * attributes cnt, hist_{0..3}s, occ_hist_{0..3}s.
    add_idx(...) scores one horizon (what the harness's per-horizon report uses); report() returns every horizon
    incl. 0 s, which the reference accumulates (:533-535) but never prints."""

    def __init__(self, save_dir='.', num_classes=18, use_lidar_mask=False, use_image_mask=False, device='cuda:0'):
        self.class_names = CLASS_NAMES
        self.save_dir, self.num_classes = save_dir, num_classes
        self.use_lidar_mask, self.use_image_mask = use_lidar_mask, use_image_mask
        self.occ_names = ['free', 'occupied']
        self.horizons = (0, 2, 4, 6)
        self.metrics = {h: Metric_mIoU(num_classes=num_classes, use_lidar_mask=use_lidar_mask,
                                       use_image_mask=use_image_mask, device=device)
                        for h in self.horizons}
        self.cnt = 0

    def __getattr__(self, name):
        # hist_0s .. hist_3s / occ_hist_0s .. occ_hist_3s as float arrays, like the reference's numpy accumulators
        for prefix, attr in (('occ_hist_', 'occ_hist'), ('hist_', 'hist')):
            if name.startswith(prefix) and name.endswith('s') and name[len(prefix):-1].isdigit():
                sec = int(name[len(prefix):-1])
                if 2 * sec in self.__dict__.get('metrics', {}):
                    return getattr(self.metrics[2 * sec], attr)
        raise AttributeError(name)

    def add_idx(self, semantics_pred_stack, semantics_gt, mask_lidar, mask_camera, idx):
        assert idx in self.metrics
        self.metrics[idx].add_batch(semantics_pred_stack[idx // 2], semantics_gt, mask_lidar, mask_camera)

    def add_batch(self, semantics_pred, semantics_gt_temp, mask_lidar_temp, mask_camera_temp):
        self.cnt += 1
        for idx in semantics_gt_temp.keys():
            self.add_idx(semantics_pred, semantics_gt_temp[idx],
                         mask_lidar_temp[idx] if mask_lidar_temp is not None else None,
                         mask_camera_temp[idx] if mask_camera_temp is not None else None, idx)

    def _miou(self, h):
        iu = Metric_mIoU.per_class_iu(self.metrics[h].hist)
        return iu, round(np.nanmean(iu[:self.num_classes - 1]) * 100, 2)

    def count_miou(self, verbose=False):
        res = []
        for sec in (1, 2, 3):
            iu, m = self._miou(2 * sec)
            if verbose:
                print('===> mIoU of %d samples at %ds: %s' % (self.cnt, sec, m))
            res.append(m)
        return self._miou(2)[0], res

    def count_iou(self):
        res = []
        for sec in (1, 2, 3):
            iu = Metric_mIoU.per_class_iu(self.metrics[2 * sec].occ_hist)
            res.append(round(iu[-1] * 100, 2))
        return res

    def report(self):
        out = {h: self._miou(h)[1] for h in self.horizons}
        out['avg_future'] = round(float(np.mean([out[h] for h in self.horizons if h != 0])), 2)
        return out
