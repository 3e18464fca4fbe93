"""Full-size composed parity (VERDICT r1 task 1): BASELINE.json configs[1] (C2), configs[2] (C3) and the literal
configs[4] render shape (C5) through the shipped entry points, stage by stage against the CPU oracle.

Every comparison PRINTS the achieved error (run with -s to see it; the same numbers are written to
gpurun_out/parity_fullsize.json) and asserts a stated bound:
  * tensors:  max|got - want| <= rtol * max|want|   (rtol per stage below; fp32 MFMA / Winograd accumulate in a
    different order than the scalar oracle),
  * occupancy grids: agreement >= 0.99999 (SURVEY section 7 asked for 0.9999; measured: <= 2 flips in 640 000), and EVERY disagreeing voxel must be a near-tie in the ORACLE's
    logits: oracle top-1 logit minus the oracle's logit of the class the GPU chose <= 2 x the MEASURED max logits error
    of that state -- i.e. no flip that the tensor tolerance does not already explain.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from preworld_amd import harness, modules as M, ops, synth as S

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GC = S.GRID_CONFIG_FULL
REPORT = {}

# stage tolerances (relative to the stage's max |value|)
# measured on MI355X (round 2, profiles/r02_parity_fullsize.json): pre 7e-7, enc 3.7e-6, neck 3.1e-6, final_conv 3.0e-6,
# forecast states 2.5e-6, logits 5.5e-6 -> bounds at 3-4x the measurement; <= 2 of 640 000 voxels flip per state
RTOL = dict(pool=0.0, pre=3e-6, enc=1.2e-5, neck=1e-5, final_conv=1e-5, state=1e-5, logits=2e-5)
AGREE_FLOOR = 0.99999
REPORT_ONLY = os.environ.get('PW_PARITY_REPORT_ONLY', '0') == '1'      # measure without asserting (tolerance setting)


def _cmp(name, got, want, rtol):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = float(np.abs(want).max())
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    REPORT[name] = dict(max_abs_err=err, max_abs=scale, rel=err / max(scale, 1e-30), bound=rtol)
    print('%-28s max|err| %.3e  max|ref| %.3e  rel %.2e  (bound %.0e)' % (name, err, scale, err / max(scale, 1e-30), rtol))
    assert REPORT_ONLY or err <= rtol * scale, (name, err, scale)
    return err


def _cmp_states(name, got_u8, want_u8, oracle_logits, logits_err):
    """got/want (X,Y,Z) uint8; oracle_logits (X,Y,Z,18)."""
    got = got_u8.cpu().numpy() if isinstance(got_u8, torch.Tensor) else got_u8
    assert got.dtype == np.uint8 and got.shape == want_u8.shape
    diff = got != want_u8
    n = int(diff.sum())
    agree = 1.0 - n / diff.size
    margin = 0.0
    if n:
        lg = oracle_logits[diff]                                              # (n,18)
        margin = float((lg.max(-1) - np.take_along_axis(lg, got[diff][:, None].astype(np.int64), 1)[:, 0]).max())
    REPORT[name] = dict(agreement=agree, flipped=n, max_oracle_margin_of_flips=margin, logits_err=logits_err)
    print('%-28s agreement %.6f  (%d of %d voxels differ; largest oracle top-2 margin among them %.3e, logits err %.3e)'
          % (name, agree, n, diff.size, margin, logits_err))
    assert REPORT_ONLY or agree >= AGREE_FLOOR, (name, agree)
    assert REPORT_ONLY or margin <= 2.0 * logits_err + 1e-7, (name, margin, logits_err)


def _cl(a):
    """oracle (B,C,Z,Y,X) -> channels-last (B,Z,Y,X,C)"""
    return np.ascontiguousarray(a.transpose(0, 2, 3, 4, 1))


_ORACLE_CACHE = {}


def _oracle_encoder(seed, sd, with_prev):
    """(cached per (seed, with_prev): the precision-parametrised tests share one oracle run)"""
    key = (seed, with_prev)
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = _oracle_encoder_now(seed, sd, with_prev)
    return _ORACLE_CACHE[key]


def _oracle_encoder_now(seed, sd, with_prev):
    bevs, pres = [], []
    for f in range(2 if with_prev else 1):
        depth, feat = S.lift_inputs(seed * 16 + f, N=6)
        r = S.synthetic_rig(6, dx=-2.5 * f)
        bev = O.lss_view_transform(depth, feat, r['sensor2ego'], r['intrin'], r['post_rot'], r['post_tran'], r['bda'],
                                   GC, S.INPUT_SIZE, S.DOWNSAMPLE)
        bevs.append(bev)
        pres.append(O.pre_process(bev, sd))
    adj = pres[1] if with_prev else np.zeros_like(pres[0])
    x = np.concatenate([adj, pres[0]], axis=1)
    feats = O.custom_resnet3d(x, sd, 'img_bev_encoder_backbone', [1, 2, 4], [1, 2, 2])
    neck = O.lss_fpn3d(feats, sd, 'img_bev_encoder_neck')
    vf = O.final_conv(neck, sd)                                                # (1,X,Y,Z,C)
    return bevs, pres, feats, neck, vf


def _gpu_stages(net, frames):
    """the same stages through the shipped modules, channels-last"""
    with torch.no_grad():
        vt = net.img_view_transformer
        _, _, size = vt._grid()
        pools = []
        for fr in frames:
            inp = [None, fr['sensor2keyego'], None, fr['intrin'], fr['post_rot'], fr['post_tran'], fr['bda']]
            inp[0] = fr['depth'].new_empty(1, 6, 1, 32, 88)
            bev, _ = vt.view_transform(inp, fr['depth'], fr['tran_feat'])
            pools.append(M.to_channels_last_3d(bev))
        pres = [net.pre_process_net.forward_cl(p)[0] for p in pools]
        adj = pres[1] if len(pres) > 1 else torch.zeros_like(pres[0])
        x = torch.cat([adj, pres[0]], dim=-1)
        feats = net.img_bev_encoder_backbone.forward_cl(x)
        neck = net.img_bev_encoder_neck.forward_cl(feats)
        fc = net.final_conv.forward_cl(neck)
    # (the composed path keeps these tensors in split-fp16 storage end to end; here every stage hands out fp32 = hi + lo
    # and the next one splits it again -- the same VALUES, but a value that sits exactly between two fp16 neighbours can
    # come back as the other (hi, lo) pair, which moves the dropped lo.lo product: last-bit differences, checked below)
    return pools, pres, feats, neck, fc


def _dump():
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_fullsize.json', 'w') as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


@pytest.fixture(params=['h2', 'f32'])
def prec(request, monkeypatch):
    """both arithmetic paths at full size: the default split-fp16 kernels and PW_PRECISION=f32 (exact-fp32
    MFMA: Winograd / direct).  The fp32 Winograd transforms lose about a bit more than the split-fp16 products (measured 1e-5 vs
    5.5e-6 relative on the logits): its bounds are twice the table's."""
    monkeypatch.setenv('PW_PRECISION', request.param)
    scale = 2.0 if request.param == 'f32' else 1.0
    saved = dict(RTOL)
    for k in RTOL:
        RTOL[k] = saved[k] * scale
    yield request.param
    RTOL.update(saved)


def _encoder_checks(tag, net, frames, sd, with_prev, seed):
    bevs, opres, ofeats, oneck, ovf = _oracle_encoder(seed, sd, with_prev)
    pools, pres, feats, neck, fc = _gpu_stages(net, frames)
    for f in range(len(bevs)):
        got = pools[f].cpu().numpy()
        assert np.array_equal(got, _cl(bevs[f])), '%s: pooled frame %d is not bit-identical to the oracle' % (tag, f)
        REPORT['%s pool frame%d' % (tag, f)] = dict(bit_exact=True, nonzero_voxels=int((np.abs(got).sum(-1) > 0).sum()))
        print('%-28s bit-exact (%d non-empty voxels)' % ('%s pool frame%d' % (tag, f), REPORT['%s pool frame%d' % (tag, f)]['nonzero_voxels']))
        _cmp('%s pre_process frame%d' % (tag, f), pres[f], _cl(opres[f]), RTOL['pre'])
    for i in range(3):
        _cmp('%s enc%d' % (tag, i), feats[i], _cl(ofeats[i]), RTOL['enc'])
    _cmp('%s neck' % tag, neck, _cl(oneck), RTOL['neck'])
    # oracle final_conv is (1,X,Y,Z,C); ours (1,Z,Y,X,C)
    _cmp('%s final_conv' % tag, fc, np.ascontiguousarray(ovf.transpose(0, 3, 2, 1, 4)), RTOL['final_conv'])
    return ovf, fc


def test_c2_single_frame_with_prev_false_fullsize(prec):
    """configs[1]: 6 cams, 200x200x16, `with_prev=False` (adjacent slice = zeros, bevdet_occ.py:243-258), the PreWorld
    detector with the OccHead decode (preworld.py:196-221), 1 state."""
    sd = S.synth_state_dict(0)
    net = harness.build_model(harness.model_cfg(GC, with_prev=False, detector='PreWorld'), sd, DEV)
    assert type(net).__name__ == 'PreWorld' and not net.with_prev
    frames = harness.lifted_frames(5, 6, DEV, n_frames=2)             # the adjacent frame is supplied and must be ignored
    TAG = 'C2 ' + prec
    ovf, fc = _encoder_checks(TAG, net, frames[:1], sd, False, 5)
    with torch.no_grad():
        res = net.simple_test_from_lift(frames, want_logits=True)
    assert sorted(k for k in res if 'occ' in k) == ['geo_occ', 'semantic_occ']
    _cmp('%s composed vs staged final_conv' % TAG, M.as_f32(res['voxel_feats'][0]), fc.cpu().numpy(), 6e-6 if prec == 'h2' else 2e-5)
    occ_o, logits_o = O.occ_decode(ovf, sd)
    lerr = _cmp(TAG + ' logits', res['logits'][0][0].permute(2, 1, 0, 3), logits_o, RTOL['logits'])
    _cmp_states(TAG + ' semantic_occ', res['semantic_occ'][0], occ_o, logits_o, lerr)
    geo = res['geo_occ'][0].cpu().numpy()
    np.testing.assert_array_equal(geo, np.where(res['semantic_occ'][0].cpu().numpy() != 17, 0, 17).astype(np.uint8))
    _dump()


def test_c3_seven_states_fullsize(prec):
    """configs[2]: key + adjacent frame, 7 states through PreWorld4DTraj.simple_test_from_lift
    (preworld_temporal_traj.py:212-370, post-finetune branch), every stage and every state against the oracle."""
    sd = S.synth_state_dict(0)
    net = harness.build_model(harness.model_cfg(GC), sd, DEV)
    frames = harness.lifted_frames(6, 6, DEV, n_frames=2)
    ego = torch.from_numpy(S.ego_state(6)).to(DEV)
    TAG = 'C3 ' + prec
    ovf, fc = _encoder_checks(TAG, net, frames, sd, True, 6)
    with torch.no_grad():
        res = net.simple_test_from_lift(frames, ego, n_steps=6, want_logits=True)
    _cmp('%s composed vs staged final_conv' % TAG, M.as_f32(res['voxel_feats'][0]), fc.cpu().numpy(), 6e-6 if prec == 'h2' else 2e-5)
    e = O.plan_head(S.ego_state(6).reshape(1, -1).astype(np.float32), sd)[0]
    v = ovf
    for k in range(7):
        if k:
            v = O.forecast_step(v, e, sd)
            _cmp(TAG + ' state %d features' % k, M.as_f32(res['voxel_feats'][k]), np.ascontiguousarray(v.transpose(0, 3, 2, 1, 4)), RTOL['state'])
        occ_o, logits_o = O.occ_decode(v, sd)
        lerr = _cmp(TAG + ' logits %ds' % k, res['logits'][k][0].permute(2, 1, 0, 3), logits_o, RTOL['logits'])
        _cmp_states(TAG + ' semantic_occ_%ds' % k, res['semantic_occ_%ds' % k][0], occ_o, logits_o, lerr)
    _dump()


class _UniformConsts(O.NerfConsts):
    """configs[4] literal sampling: S uniform midpoints in t in (0, 2) instead of nerf_head.py:35-43's 391 + 26"""

    def __init__(self, n):
        super().__init__()
        b = torch.linspace(0, 2, n + 1).numpy()
        self._t = ((b[1:] + b[:-1]) * np.float32(0.5)).astype(np.float32)

    def t_table(self):
        return self._t


def test_c5_literal_shape_3072_rays_x_96_samples():
    """configs[4]: 6 x 512 = 3072 rays x 96 uniform samples through pw_render_rays against the oracle."""
    head = M.NerfHead(point_cloud_range=[-40, -40, -1, 40, 40, 5.4], voxel_size=0.4, scene_center=[0, 0, 2.2], radius=39,
                      use_depth_sup=True).to(DEV)
    R, NS = 3072, 96
    density, semantic, color = S.render_grids(41)
    o, d = S.rays(42, R)
    bda = np.eye(3, dtype=np.float32)
    consts = _UniformConsts(NS)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)     # noqa: E731
    grid = M.pack_attribute_grid(T(density), T(semantic), T(color))
    out = ops.render_rays(T(o), T(d), T(consts.t_table()), grid, head.consts(torch.from_numpy(bda)), want_debug=True)
    res = O.render_one_scene(o, d, bda, density, semantic, color, consts)
    depth, sem, col = O.render_outputs(res, consts)
    m = (out['mask'].cpu().numpy() == res['sample_mask']).mean()
    print('C5 3072x96 sample-mask agreement %.6f, kept samples %d (oracle %d)' % (m, int(out['counts'][:, 2].sum()), len(res['weights'])))
    assert m > 0.9999
    for name, got, want, tol in (('alphainv_last', out['alphainv_last'], res['alphainv_last'], 2e-6),
                                 ('depth', out['depth'], depth, 2e-6), ('semantic', out['semantic'], sem, 2e-6),
                                 ('color', out['color'], col, 2e-6)):
        _cmp('C5 3072x96 ' + name, got, want.astype(np.float32), tol)
    _dump()
