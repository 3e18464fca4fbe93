"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (preworld_amd/).

Differentiable restatement of the reference's render head for gradient parity: NerfHead.render_one_scene +
render_depth / render_semantic / render_color (mmdet3d/models/nerf/nerf_head.py:165-269, 331-353) with torch-CPU autograd.
F.grid_sample is torch's own (the reference calls the same function, :213-225); the two native ops with a backward
(Raw2Alpha, Alphas2Weights -- mmdet3d/models/nerf/utils.py:26-68) are autograd Functions over the C oracle's restatements of
render_utils_kernel.cu:431-443,507-517,577-677; sampling and the cumdist mask (no gradient) come from the numpy oracle.
Pinned by tools/gen_golden.py `render_grad`: outputs AND gradients equal the imported reference NerfHead's (fixture
tests/golden/render_grad_small.npz)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O


class Raw2Alpha(torch.autograd.Function):
    @staticmethod
    def forward(ctx, density, shift, interval):
        e, a = O.raw2alpha(density.detach().numpy(), float(shift), float(interval))
        ctx.save_for_backward(torch.from_numpy(e))
        ctx.interval = float(interval)
        return torch.from_numpy(a)

    @staticmethod
    def backward(ctx, grad_back):
        (e,) = ctx.saved_tensors
        return torch.from_numpy(O.raw2alpha_backward(e.numpy(), grad_back.contiguous().numpy(), ctx.interval)), None, None


class Alphas2Weights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, ray_id, N):
        w, T, last, i_s, i_e = O.alpha2weight(alpha.detach().numpy(), ray_id.numpy(), int(N))
        ctx.save_for_backward(alpha.detach(), *[torch.from_numpy(x) for x in (w, T, last, i_s, i_e)])
        ctx.n_rays = int(N)
        return torch.from_numpy(w), torch.from_numpy(last)

    @staticmethod
    def backward(ctx, grad_weights, grad_last):
        alpha, w, T, last, i_s, i_e = ctx.saved_tensors
        g = O.alpha2weight_backward(alpha.numpy(), w.numpy(), T.numpy(), last.numpy(), i_s.numpy(), i_e.numpy(), ctx.n_rays,
                                    grad_weights.contiguous().numpy(), grad_last.contiguous().numpy())
        return torch.from_numpy(g), None, None


def render(rays_o, rays_d, bda, density, semantic, color, consts=None):
    """rays (R,3) numpy; bda (3,3) numpy; density (X,Y,Z), semantic (X,Y,Z,17), color (X,Y,Z,3) torch tensors (may
    require grad).  Returns dict(depth (R), semantic (R,17), color (R,3), alphainv_last (R), weights (R,S) dense)."""
    consts = consts or O.NerfConsts()
    pts, inner, t = O.sample_ray(rays_o, rays_d, consts, bda)
    R, S = inner.shape
    mask = inner.copy().astype(bool)
    dist_thres = np.float32((2 + 2 * float(consts.bg_len)) / consts.world_len * consts.step_size * 0.95)
    d = pts[:, 1:] - pts[:, :-1]
    dist = np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)
    mask[:, 1:] |= O.cumdist_thres(dist, dist_thres).astype(bool)
    ray_id = torch.from_numpy(np.repeat(np.arange(R), S).reshape(R, S)[mask])
    step_id = torch.from_numpy(np.tile(np.arange(S), R).reshape(R, S)[mask])
    tt = torch.from_numpy(np.broadcast_to(t[None], (R, S))[mask].copy())
    xyz = torch.from_numpy(pts[mask]).reshape(1, 1, 1, -1, 3)
    xyz_min, xyz_max = torch.from_numpy(consts.xyz_min), torch.from_numpy(consts.xyz_max)
    ind_norm = ((xyz - xyz_min) / (xyz_max - xyz_min)).flip((-1,)) * 2 - 1
    dens = F.grid_sample(density.unsqueeze(0).unsqueeze(1), ind_norm, mode='bilinear', align_corners=True).reshape(1, -1).T.reshape(-1)
    sem = F.grid_sample(semantic.permute(3, 0, 1, 2).unsqueeze(0), ind_norm, mode='bilinear', align_corners=True)
    sem = sem.reshape(semantic.shape[-1], -1).T
    col = F.grid_sample(color.permute(3, 0, 1, 2).unsqueeze(0), ind_norm, mode='bilinear', align_corners=True).reshape(3, -1).T
    alpha = Raw2Alpha.apply(dens, float(consts.act_shift), 0.5)
    m1 = alpha > consts.fast_color_thres
    ray_id, step_id, tt, alpha, sem, col = ray_id[m1], step_id[m1], tt[m1], alpha[m1], sem[m1], col[m1]
    weights, last = Alphas2Weights.apply(alpha, ray_id, R)
    m2 = weights > consts.fast_color_thres
    ray_id, step_id, tt, weights, sem, col = ray_id[m2], step_id[m2], tt[m2], weights[m2], sem[m2], col[m2]
    s = 1 - 1 / (1 + tt)
    depth = (torch.zeros(R).index_add_(0, ray_id, weights * s) + 1e-7) * float(consts.radius)
    out_sem = torch.zeros(R, sem.shape[1]).index_add_(0, ray_id, weights.unsqueeze(-1) * sem)
    out_col = torch.zeros(R, 3).index_add_(0, ray_id, weights.unsqueeze(-1) * col)
    dense = torch.zeros(R, S).index_put((ray_id, step_id), weights)
    return dict(depth=depth, semantic=out_sem, color=out_col, alphainv_last=last, weights=dense)


def scalar_objective(out, coef):
    """a fixed linear functional of every differentiable output (coef: dict of tensors of the outputs' shapes)"""
    return sum((out[k] * coef[k]).sum() for k in ('depth', 'semantic', 'color', 'alphainv_last', 'weights'))


def objective_coefficients(seed, R, S):
    g = torch.Generator().manual_seed(seed)
    return dict(depth=torch.randn(R, generator=g) * 0.1, semantic=torch.randn(R, 17, generator=g),
                color=torch.randn(R, 3, generator=g), alphainv_last=torch.randn(R, generator=g),
                weights=torch.randn(R, S, generator=g))


def nerf_head_forward(density, semantic, color, rays, bda, class_weights, if_temporal=False, interval=0, weight_entropy_last=0.01,
                      weight_distortion=0.01):
    """differentiable NerfHead.forward (nerf_head.py:355-420, compute_loss[_temporal] :271-329, silog_loss / l1_loss nerf/utils.py:71-87,
    the distortion term as oracle.flatten_eff_distloss restates the absent torch_efficient_distloss) on top of render(): torch tensors
    density (B,X,Y,Z), semantic (B,X,Y,Z,17), color (B,X,Y,Z,3) that may require grad; rays (B,R,16) / bda (B,3,3) numpy.  All loss weights
    1 except the two given.  Pinned by tests/golden/nerf_losses_small.npz: loss values AND grid gradients of the imported reference."""
    sfx = '_%ds' % int(interval) if if_temporal else ''
    cw = torch.from_numpy(np.asarray(class_weights)).float()
    t_tab = torch.from_numpy(O.NerfConsts().t_table())
    s_tab = 1 - 1 / (1 + t_tab)
    losses = {}
    for b in range(rays.shape[0]):
        gt_depth = rays[b, :, 2]
        gt_depth[gt_depth > 52] = 0
        m = gt_depth > 0
        out = render(np.ascontiguousarray(rays[b, :, 4:7][m]), np.ascontiguousarray(rays[b, :, 7:10][m]), bda[b], density[b], semantic[b], color[b])
        tgt_d, tgt_s, tgt_c = [torch.from_numpy(np.ascontiguousarray(a)) for a in (gt_depth[m], rays[b, :, 3][m], rays[b, :, 13:16][m])]
        d = torch.log(out['depth'] + 1e-7) - torch.log(tgt_d)
        single = {'loss_render_depth': torch.sqrt((d ** 2).mean() - 0.85 * d.mean() ** 2),
                  'loss_render_semantic': F.cross_entropy(out['semantic'], tgt_s.long(), weight=cw),
                  'loss_render_color': torch.sum(torch.mean(torch.abs(out['color'] - tgt_c), dim=0))}
        if weight_entropy_last > 0:
            p = out['alphainv_last'].clamp(1e-6, 1 - 1e-6)
            single['loss_sdf_entropy'] = weight_entropy_last * -(p * torch.log(p) + (1 - p) * torch.log(1 - p)).mean()
        if weight_distortion > 0:
            w = out['weights']                                            # (R,S) dense, 0 where culled: kept <=> w > 1e-7
            kept = w > 0
            n_max = int(kept.sum())
            ray_of = torch.nonzero(kept.any(1)).reshape(-1)
            n_rays = int(ray_of.max()) + 1 if len(ray_of) else 1
            w_pre = torch.cumsum(w, 1) - w
            wm_pre = torch.cumsum(w * s_tab, 1) - w * s_tab
            single['loss_sdf_distortion'] = weight_distortion * ((1 / 3) * (1 / max(n_max, 1)) * w.pow(2)
                                                                   + 2 * w * (s_tab * w_pre - wm_pre)).sum() / n_rays
        for k, v in single.items():
            losses[k + sfx] = losses[k + sfx] + v if k + sfx in losses else v
    return {k: v / semantic.shape[0] for k, v in losses.items()}
