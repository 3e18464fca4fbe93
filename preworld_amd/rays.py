"""GPU drop-in for mmdet3d/datasets/ray.py:34-119: the (R,16) ray table and the
weighted-ray-sampling weights of the pre-train dataloader.  Same call signature as the reference's
`generate_rays`; inputs are device tensors.  The draw itself (WeightedRandomSampler without
replacement, ray.py:116-118) is torch.multinomial on the device -- same distribution, not the same
random stream."""
import torch

from . import ops


def pts2ray(coor, label_depth, label_seg, label_img, c2w, cam_intrinsic):
    return ops.pts2ray(coor, label_depth, label_seg, label_img, c2w, cam_intrinsic)


def generate_rays(coors, label_depths, label_segs, label_imgs, c2w, intrins, max_ray_nums=0, time_ids=None,
                  dynamic_class=None, balance_weight=None, weight_adj=0.3, weight_dyn=0.0, use_wrs=True,
                  return_weights=False, generator=None):
    rays, ids = [], []
    for time_id in time_ids:                       # frames
        for i in time_ids[time_id]:                # cameras of one frame
            rays.append(ops.pts2ray(coors[i], label_depths[i], label_segs[i], label_imgs[i], c2w[i], intrins[i]))
            ids.append(time_id)
    if not use_wrs:
        return torch.cat(rays, dim=0)
    dev = rays[0].device
    if balance_weight is None:                     # batch statistics (ray.py:93-97)
        counts = torch.zeros(17, device=dev, dtype=torch.int64)
        for r in rays:
            ops.class_count(r, 17, counts)
        class_nums = counts.float()
        balance_weight = torch.exp(0.005 * (class_nums.max() / class_nums - 1))
    if dynamic_class is None:
        dynamic_class = torch.zeros(0, dtype=torch.int32)
    weights = [ops.wrs_weights(r, ids[k], balance_weight.to(dev), dynamic_class, weight_adj, weight_dyn)
               for k, r in enumerate(rays)]
    rays = torch.cat(rays, dim=0)
    weights = torch.cat(weights, dim=0)
    if max_ray_nums != 0 and rays.shape[0] > max_ray_nums:
        sel = torch.multinomial(weights, max_ray_nums, replacement=False, generator=generator)
        rays = rays[sel]
        if return_weights:
            return rays, weights, sel
    return (rays, weights, None) if return_weights else rays
