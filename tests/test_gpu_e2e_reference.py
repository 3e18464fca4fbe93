"""The drop-in detectors (preworld_amd.detectors, HIP path) against tests/golden/e2e_small.npz -- outputs of the REFERENCE'S OWN
PreWorld4DTraj / PreWorld classes running end to end under tools/gen_golden.py's shim (see tests/test_e2e_reference_cpu.py for
what the fixture pins).  Both sides replace the image side by the same seeded stand-ins (tests/_e2e_stub.py) and are driven
through the reference's entry point `simple_test(points, img_metas, img=img_inputs, temporal_ego_states=...)`."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _e2e_stub as E  # noqa: E402
from preworld_amd import harness  # noqa: E402
from preworld_amd import synth as S  # noqa: E402
from preworld_amd.modules import as_f32  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
_GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
GOLD = np.load(os.path.join(_GDIR, 'e2e_small.npz'))
GOLD_C6 = np.load(os.path.join(_GDIR, 'e2e_c6.npz'))
GOLD_FULL = np.load(os.path.join(_GDIR, 'e2e_full.npz'))
GOLD_TRAIN = np.load(os.path.join(_GDIR, 'e2e_train_small.npz'))
GOLD_TRAIN_FULL = np.load(os.path.join(_GDIR, 'e2e_train_full.npz'))
GOLD_TRAIN_B2 = np.load(os.path.join(_GDIR, 'e2e_train_small_b2.npz'))
GOLD_PRETRAIN = np.load(os.path.join(_GDIR, 'e2e_pretrain_small.npz'))


def _build(det, post_ft, with_prev, variant='small'):
    net = harness.build_model(E.model_cfg(det, post_ft, with_prev, variant=variant), S.synth_state_dict(0), DEV)
    return net, E.install_image_side(net, seed=0, variant=variant)


def test_prepare_inputs_matches_reference():
    net, _ = _build('PreWorld4DTraj', True, True)
    prep = net.prepare_inputs(tuple(t.to(DEV) for t in E.img_inputs(0)), stereo=True)
    np.testing.assert_allclose(torch.stack(prep[1], 0).cpu().numpy(), GOLD['prep_sensor2keyego'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(torch.stack(prep[7][:2], 0).cpu().numpy(), GOLD['prep_curr2adjsensor'], rtol=0, atol=2e-6)
    assert prep[7][2] is None and len(prep[0]) == 3 and tuple(prep[0][0].shape) == (1, E.N_CAMS, 3) + E.input_size('small')


@pytest.mark.parametrize('tag,det,post_ft,with_prev', E.RUNS)
def test_dropin_detectors_match_reference_detectors(tag, det, post_ft, with_prev):
    """the drop-in's simple_test against the reference classes' own (e2e_small.npz): keys, every grid; a differing voxel of a
    post-finetune semantic grid must be a near-tie of the REFERENCE's logits (round 4), threshold-decode grids allow exact ties"""
    net, dn = _build(det, post_ft, with_prev)
    inputs = tuple(t.to(DEV) for t in E.img_inputs(0))
    ego = [[t.to(DEV) for t in E.ego_states(0)[0]]]
    with torch.no_grad():
        res = net.simple_test(None, None, img=inputs, temporal_ego_states=ego)
    assert dn.k == int(GOLD[tag + '_n_depthnet_calls'])
    if tag == 'p4d_ft':
        np.testing.assert_allclose(torch.stack(dn.mlp_inputs, 0).numpy(), GOLD['mlp_input'], rtol=1e-6, atol=1e-6)
    assert sorted(res.keys()) == list(GOLD[tag + '_keys'])
    for k in res:
        want = GOLD[tag + '_' + k]
        got = res[k][0]
        assert isinstance(got, np.ndarray) and got.dtype == np.uint8 and got.shape == want.shape, (k, type(got))
        flips = np.nonzero((got != want).reshape(-1))[0]
        print('[e2e] %-14s %-16s flips vs the reference classes: %d / %d' % (tag, k, flips.size, want.size))
        if post_ft and k.startswith('semantic_occ'):
            # every differing voxel must be one of the reference's OWN near-ties (the fixture lists the voxels whose top-2 logit margin
            # is below 1e-3 of the largest logit, with margin and runner-up class): we chose the runner-up, and the margin is within
            # twice the logit error bound of DESIGN.md section 6 (2e-5 of the largest logit)
            ti, tm, tc = GOLD['%s_%s_tie_idx' % (tag, k)], GOLD['%s_%s_tie_margin' % (tag, k)], GOLD['%s_%s_tie_cls' % (tag, k)]
            tol = 2 * 2e-5 * float(GOLD['%s_%s_logit_absmax' % (tag, k)])
            for v in flips:
                j = np.nonzero(ti == v)[0]
                assert j.size == 1, (tag, k, int(v), 'differs at a voxel that is not a near-tie of the reference logits')
                assert got.reshape(-1)[v] == tc[j[0]] and tm[j[0]] <= tol, (tag, k, int(v), float(tm[j[0]]), tol)
        else:
            assert flips.size <= 3, (tag, k, flips.size)               # threshold decode / geo grids: exact ties only
    # intermediate tensors: encoder output and voxel_feats at the sampled voxels
    dn.reset()
    with torch.no_grad():
        frames = net.lift_inputs_from_images(net.prepare_inputs(inputs, stereo=True))
        bev = net._ranged(lambda: as_f32(net.extract_bev_feat_cl(frames)))
        dn.reset()
        vf = net._ranged(lambda: as_f32(net.extract_voxel_feat_cl(frames)))
    idx = torch.from_numpy(GOLD['sample_idx']).to(DEV)
    for name, t in (('bev', bev), ('vf', vf)):
        rows = t[0].reshape(-1, 32)[idx].cpu().numpy()                            # channels-last (Z,Y,X,C): row = z*Y*X + y*X + x
        want = GOLD['%s_%s_rows' % (tag, name)]
        err = float(np.abs(rows - want).max()) / float(np.abs(want).max())
        print('[e2e] %-14s %-4s rows: max|err| / max|ref| = %.2e' % (tag, name, err))
        assert err <= 1e-5, (tag, name, err)


@pytest.mark.parametrize('variant,shape', [('c6', (100, 100, 8)), ('full', (200, 200, 16))])
def test_dropin_detector_matches_reference_detector_six_cameras(variant, shape):
    """the same comparison with the full rig: on BASELINE.json configs[0]'s grid (6 cameras, 100 x 100 x 8; tests/golden/e2e_c6.npz)
    and on the HEADLINE config itself (200 x 200 x 16, 7 states, 512 x 1408 image = 1 486 848 frustum points per frame; e2e_full.npz,
    round 5: the round-4 fixture had a 128 x 352 image, 1/16 of the lift) -- the reference's own
    PreWorld4DTraj.simple_test vs the drop-in, every differing voxel explained by the reference's logits; encoder output and
    voxel_feats rows at 1 024 sampled voxels to 1e-5 of the largest value."""
    G, tag = (GOLD_C6 if variant == 'c6' else GOLD_FULL), 'p4d_ft'
    net, dn = _build('PreWorld4DTraj', True, True, variant=variant)
    inputs = tuple(t.to(DEV) for t in E.img_inputs(0, variant))
    ego = [[t.to(DEV) for t in E.ego_states(0)[0]]]
    with torch.no_grad():
        res = net.simple_test(None, None, img=inputs, temporal_ego_states=ego)
    assert dn.k == int(G[tag + '_n_depthnet_calls']) and sorted(res.keys()) == list(G[tag + '_keys'])
    np.testing.assert_allclose(torch.stack(dn.mlp_inputs, 0).numpy(), G['mlp_input'], rtol=1e-6, atol=1e-6)
    n_flips = 0
    for k in res:
        got = res[k][0]
        if tag + '_' + k in G.files:
            want = G[tag + '_' + k]
        else:                                                   # geo grids of the full-size fixture: {0, 17}-valued, stored as bits
            want = (np.unpackbits(G[tag + '_' + k + '_is17_bits'])[:got.size].reshape(got.shape) * 17).astype(np.uint8)
        assert got.dtype == np.uint8 and got.shape == want.shape == shape
        flips = np.nonzero((got != want).reshape(-1))[0]
        n_flips += flips.size
        print('[e2e %s] %-16s flips vs the reference class: %d / %d' % (variant, k, flips.size, want.size))
        if k.startswith('semantic_occ'):
            ti, tm, tc = G['%s_%s_tie_idx' % (tag, k)], G['%s_%s_tie_margin' % (tag, k)], G['%s_%s_tie_cls' % (tag, k)]
            tol = 2 * 2e-5 * float(G['%s_%s_logit_absmax' % (tag, k)])
            for v in flips:
                j = np.nonzero(ti == v)[0]
                assert j.size == 1 and got.reshape(-1)[v] == tc[j[0]] and tm[j[0]] <= tol, (k, int(v))
        else:
            # geo grids (preworld_temporal_traj.py:315-322): 0 where the semantic grid is occupied, 17 elsewhere -- exactly the function of
            # OUR semantic grid of the same state, whose every difference from the reference's is explained above
            sem = res[k.replace('geo_occ', 'semantic_occ')][0]
            np.testing.assert_array_equal(got, np.where(sem != 17, 0, 17).astype(np.uint8))
            want_sem = G[tag + '_' + k.replace('geo_occ', 'semantic_occ')]
            assert flips.size <= int((sem != want_sem).sum())           # a geo voxel differs only where the semantic voxel does
    assert n_flips <= 2e-5 * 14 * np.prod(shape), n_flips
    dn.reset()
    with torch.no_grad():
        frames = net.lift_inputs_from_images(net.prepare_inputs(inputs, stereo=True))
        bev = net._ranged(lambda: as_f32(net.extract_bev_feat_cl(frames)))
        dn.reset()
        vf = net._ranged(lambda: as_f32(net.extract_voxel_feat_cl(frames)))
    idx = torch.from_numpy(G['sample_idx']).to(DEV)
    for name, t in (('bev', bev), ('vf', vf)):
        rows, want = t[0].reshape(-1, 32)[idx].cpu().numpy(), G['%s_%s_rows' % (tag, name)]
        err = float(np.abs(rows - want).max()) / float(np.abs(want).max())
        print('[e2e %s] %-4s rows: max|err| / max|ref| = %.2e' % (variant, name, err))
        assert err <= 1e-5, (variant, name, err)


@pytest.mark.parametrize('tag,det,variant', [('pw', 'PreWorld', 'small'), ('p4d', 'PreWorld4DTraj', 'small'), ('pw', 'PreWorld', 'full')])
def test_dropin_forward_train_matches_reference_forward_train(tag, det, variant):
    """VERDICT r03 missing 3 / next 6: the training composition against the reference's OWN forward_train (preworld.py:229-309,
    preworld_temporal_traj.py:372-530; tests/golden/e2e_train_small.npz, produced by tools/gen_golden.py gen_e2e_train running those
    methods in train() mode): the loss dict's keys (which state gets which term, `..._{k}s`), every loss value to 1e-4, and
    d sum(losses) / d of final_conv, OccHead, encoder, pre_process and (temporal) the forecast / trajectory heads' weights.
    variant 'full': PreWorld.forward_train at the HEADLINE config, 6 cameras, 200 x 200 x 16, 512 x 1408 image (32 x 88 feature map,
    1.49 M frustum points; e2e_train_full.npz, round 5)."""
    G = GOLD_TRAIN if variant == 'small' else GOLD_TRAIN_FULL
    cfg = E.model_cfg(det, True, True, variant=variant)
    cfg.update(E.TRAIN_CFG)
    net = harness.build_model(cfg, S.synth_state_dict(0), DEV).train()
    if hasattr(net, 'set_epoch'):
        net.set_epoch(E.TRAIN_EPOCH)
    E.install_image_side(net, seed=0, variant=variant)
    inputs = tuple(t.to(DEV) for t in E.img_inputs(0, variant))
    # every maximum a BatchNorm kernel recorded for the split-fp16 weight gradient is compared with a fresh pw_absmax2 pass
    from preworld_amd import train
    train._AMAX_CHECK, train._AMAX_STATS['checked'] = variant == 'small', 0
    losses = net(return_loss=True, img_inputs=inputs, img_metas=[dict()], **E.train_kwargs(0, det, DEV, variant))
    assert sorted(losses.keys()) == list(G[tag + '_keys']), sorted(losses.keys())
    for k, v in losses.items():
        want = float(G['%s_%s' % (tag, k)])
        print('[e2e train] %-4s %-24s %.7f   reference forward_train %.7f' % (tag, k, float(v), want))
        assert abs(float(v) - want) <= 1e-4 * max(abs(want), 1.0), (k, float(v), want)
    total = sum(losses.values())
    assert abs(float(total) - float(G[tag + '_total'])) <= 1e-4 * abs(float(G[tag + '_total']))
    total.backward()
    if variant == 'small':
        assert train._AMAX_STATS['checked'] > 0
    train._AMAX_CHECK = False
    for name, p in E.grad_probes(net, det):
        g = p.grad.detach().reshape(-1)
        want = G['%s_grad_%s' % (tag, name)]
        got = (g if g.numel() <= 40000 else g[::7]).cpu().numpy()
        d = got.astype(np.float64) - want
        l2, mx = float(np.linalg.norm(d) / np.linalg.norm(want)), float(np.abs(d).max() / np.abs(want).max())
        nrm = float(p.grad.double().norm()) / float(G['%s_gradnorm_%s' % (tag, name)])
        print('[e2e train] %-4s d / d %-18s l2 err / l2 %.2e   max err / max %.2e   |g| / |g_ref| %.6f' % (tag, name, l2, mx, nrm))
        # (ReLU units whose pre-activation is within fp32 rounding of zero fall on different sides: tests/test_gpu_train.py)
        assert l2 <= 1e-2 and mx <= 3e-2 and abs(nrm - 1.0) <= 2e-3, (name, l2, mx, nrm)
    bn = net.occupancy_head.occ_convs[0][1]
    assert int(bn.num_batches_tracked) == int(G[tag + '_occ_bn_batches'])
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), G[tag + '_occ_bn_running_mean'], rtol=1e-4, atol=1e-5)


def _check_forward_train(G, tag, det, net, losses, probes, loss_tol=1e-4):
    """losses and gradients of one drop-in forward_train against a fixture written by tools/gen_golden.py _run_reference_forward_train"""
    assert sorted(losses.keys()) == list(G[tag + '_keys']), sorted(losses.keys())
    for k, v in losses.items():
        want = float(G['%s_%s' % (tag, k)])
        print('[e2e train] %-4s %-28s %.7f   reference forward_train %.7f' % (tag, k, float(v), want))
        assert abs(float(v) - want) <= loss_tol * max(abs(want), 1.0 if 'sdf' not in k else 1e-3), (k, float(v), want)
    total = sum(losses.values())
    assert abs(float(total) - float(G[tag + '_total'])) <= loss_tol * abs(float(G[tag + '_total']))
    total.backward()
    for name, p in probes:
        want = G['%s_grad_%s' % (tag, name)]
        if float(G['%s_gradnorm_%s' % (tag, name)]) == 0.0:           # reached through a zero-weight term only: exactly zero
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            print('[e2e train] %-4s d / d %-18s exactly zero on both sides' % (tag, name))
            continue
        g = p.grad.detach().reshape(-1)
        got = (g if g.numel() <= 40000 else g[::7]).cpu().numpy()
        d = got.astype(np.float64) - want
        l2, mx = float(np.linalg.norm(d) / np.linalg.norm(want)), float(np.abs(d).max() / np.abs(want).max())
        nrm = float(p.grad.double().norm()) / float(G['%s_gradnorm_%s' % (tag, name)])
        print('[e2e train] %-4s d / d %-18s l2 err / l2 %.2e   max err / max %.2e   |g| / |g_ref| %.6f' % (tag, name, l2, mx, nrm))
        assert l2 <= 1e-2 and mx <= 3e-2 and abs(nrm - 1.0) <= 2e-3, (name, l2, mx, nrm)
    for name, bn in (('occ', net.occupancy_head.occ_convs[0][1]), ('enc', net.img_bev_encoder_backbone.layers[0][0].conv1.bn),
                     ('pre', net.pre_process_net.layers[0][0].conv1.bn)):
        assert int(bn.num_batches_tracked) == int(G['%s_%s_bn_batches' % (tag, name)]), name
        np.testing.assert_allclose(bn.running_mean.cpu().numpy(), G['%s_%s_bn_running_mean' % (tag, name)], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), G['%s_%s_bn_running_var' % (tag, name)], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('tag,det', [('pw', 'PreWorld'), ('p4d', 'PreWorld4DTraj')])
def test_dropin_forward_train_batch2_matches_reference(tag, det):
    """the fine-tune forward_train at the reference's training batch, samples_per_gpu = 2
    (configs/preworld/nuscenes/preworld-7frame-finetune.py:58; e2e_train_small_b2.npz = the reference's own forward_train on two collated
    samples): BatchNorm statistics over both samples (running mean AND variance of an OccHead, an encoder and a pre_process layer),
    the per-batch OccHead loop (preworld.py:240-247), (B,X,Y,Z) labels, per-sample ego states / trajectories."""
    cfg = E.model_cfg(det, True, True)
    cfg.update(E.TRAIN_CFG)
    net = harness.build_model(cfg, S.synth_state_dict(0), DEV).train()
    if hasattr(net, 'set_epoch'):
        net.set_epoch(E.TRAIN_EPOCH)
    E.install_image_side(net, seed=0)
    inputs = tuple(t.to(DEV) for t in E.img_inputs(0, batch=2))
    losses = net(return_loss=True, img_inputs=inputs, img_metas=[dict(), dict()], **E.train_kwargs(0, det, DEV, batch=2))
    _check_forward_train(GOLD_TRAIN_B2, tag, det, net, losses, E.grad_probes(net, det))


@pytest.mark.parametrize('tag,det', [('pw', 'PreWorld'), ('p4d', 'PreWorld4DTraj')])
def test_dropin_forward_train_pretrain_matches_reference(tag, det):
    """VERDICT r05 item 1b (BASELINE configs[4]'s composition): forward_train under the PRE-TRAIN flags of
    configs/preworld/nuscenes/preworld-7frame-pretrain.py:10-33 and nuscenes-temporal/preworld-7frame-pretrain-traj.py -- if_render=True,
    if_post_finetune=False, render weights 1 / 1 / 1 / 0.01 / 0.01, LSS depth loss on for PreWorld -- against the reference's OWN
    PreWorld.forward_train / PreWorld4DTraj.forward_train (e2e_pretrain_small.npz), B = 2, epoch 4 (temporal_rays[1..3] on the forecast
    states): the zero-weight loss_sup_voxel (preworld.py:130-135), the render losses of every supervised state under their `_{k}s` keys,
    loss_lss_depth, trajectory terms; gradients of final_conv, all three attribute MLPs, encoder, pre_process, forecast heads; the
    OccHead's gradient exactly zero.  A good part of the rays terminate (fixture `*_render_stats`)."""
    G = GOLD_PRETRAIN
    cfg = E.model_cfg(det, True, True)
    cfg.update(E.pretrain_cfg(det))
    net = harness.build_model(cfg, E.opaque_density_state(S.synth_state_dict(0)), DEV).train()
    if hasattr(net, 'set_epoch'):
        net.set_epoch(E.PRETRAIN_EPOCH)
    E.install_image_side(net, seed=0)
    inputs = tuple(t.to(DEV) for t in E.img_inputs(0, batch=2))
    st = G[tag + '_render_stats']
    print('[e2e pretrain] %s: reference (rays, terminated, partly opaque) per NerfHead batch element: %s' % (tag, st.tolist()))
    assert st[:, 1].sum() >= 10 and st[:, 2].sum() >= 100
    losses = net(return_loss=True, img_inputs=inputs, img_metas=[dict(), dict()], **E.pretrain_kwargs(0, det, DEV, batch=2))
    _check_forward_train(G, tag, det, net, losses, E.pretrain_grad_probes(net, det), loss_tol=3e-4)


def test_simple_test_with_capture_replay_matches_the_reference_and_the_eager_entry():
    """round 6: `net.capture_replay = True` routes the reference entry point simple_test() through the module's own hipGraph of the
    hot path (PreWorld4DTraj.simple_test_captured).  Same keys, numpy uint8 payload; against the reference classes' grids
    (e2e_small.npz) under the same near-tie rule as the eager entry; a second, different sample through the SAME captured graph
    (inputs copied into its static buffers, ranges checked on the host) equals the eager result of that sample up to exact ties."""
    tag = 'p4d_ft'
    net, dn = _build('PreWorld4DTraj', True, True)
    inputs = tuple(t.to(DEV) for t in E.img_inputs(0))
    ego = [[t.to(DEV) for t in E.ego_states(0)[0]]]
    net.capture_replay = True
    with torch.no_grad():
        res = net.simple_test(None, None, img=inputs, temporal_ego_states=ego)
    assert sorted(res.keys()) == list(GOLD[tag + '_keys']) and len(net._captured) == 1
    for k in res:
        want, got = GOLD[tag + '_' + k], res[k][0]
        assert isinstance(got, np.ndarray) and got.dtype == np.uint8 and got.shape == want.shape
        flips = np.nonzero((got != want).reshape(-1))[0]
        if k.startswith('semantic_occ'):
            ti, tm, tc = GOLD['%s_%s_tie_idx' % (tag, k)], GOLD['%s_%s_tie_margin' % (tag, k)], GOLD['%s_%s_tie_cls' % (tag, k)]
            tol = 2 * 2e-5 * float(GOLD['%s_%s_logit_absmax' % (tag, k)])
            for v in flips:
                j = np.nonzero(ti == v)[0]
                assert j.size == 1 and got.reshape(-1)[v] == tc[j[0]] and tm[j[0]] <= tol, (k, int(v))
        else:
            assert flips.size <= 3, (k, flips.size)
    # another sample (second batch element's poses, its own seeded DepthNet outputs) through the same graph vs the eager entry
    inputs2 = tuple(t.to(DEV) for t in E._img_inputs_one(0, 'small', 1))
    ego2 = [[torch.from_numpy(S.ego_state(47)).to(DEV)]]
    dn.reset()
    dn.seed = 3
    with torch.no_grad():
        cap = net.simple_test(None, None, img=inputs2, temporal_ego_states=ego2)
    assert len(net._captured) == 1                                        # the same shapes: replayed, not re-captured
    net.capture_replay = False
    dn.reset()
    with torch.no_grad():
        eager = net.simple_test(None, None, img=inputs2, temporal_ego_states=ego2)
    n_diff = sum(int((cap[k][0] != eager[k][0]).sum()) for k in eager)
    print('[e2e] captured vs eager simple_test on a second sample: %d of %d voxels differ' % (n_diff, sum(v[0].size for v in eager.values())))
    assert sorted(cap) == sorted(eager) and n_diff <= 8
    assert any(int((cap[k][0] != res[k][0]).sum()) > 100 for k in cap)      # it really was another sample


def test_preworld_simple_test_with_capture_replay_matches_the_reference():
    """the single-time-step detector through the same opt-in (PreWorld.simple_test -> simple_test_captured, no ego states, 1 state)
    against the reference class's grids (e2e_small.npz, run `pw_ft`)"""
    tag = 'pw_ft'
    net, dn = _build('PreWorld', True, True)
    inputs = tuple(t.to(DEV) for t in E.img_inputs(0))
    net.capture_replay = True
    with torch.no_grad():
        res = net.simple_test(None, None, img=inputs)
        dn.reset()
        res2 = net.simple_test(None, None, img=inputs)              # second call: replay
    assert sorted(res.keys()) == list(GOLD[tag + '_keys']) and len(net._captured) == 1
    for k in res:
        want, got = GOLD[tag + '_' + k], res[k][0]
        assert got.dtype == np.uint8 and got.shape == want.shape and np.array_equal(got, res2[k][0])
        flips = np.nonzero((got != want).reshape(-1))[0]
        if k.startswith('semantic_occ'):
            ti, tm, tc = GOLD['%s_%s_tie_idx' % (tag, k)], GOLD['%s_%s_tie_margin' % (tag, k)], GOLD['%s_%s_tie_cls' % (tag, k)]
            tol = 2 * 2e-5 * float(GOLD['%s_%s_logit_absmax' % (tag, k)])
            for v in flips:
                j = np.nonzero(ti == v)[0]
                assert j.size == 1 and got.reshape(-1)[v] == tc[j[0]] and tm[j[0]] <= tol, (k, int(v))
        else:
            assert flips.size <= 3, (k, flips.size)


@pytest.mark.parametrize('tag,det', [('p4d_ft', 'PreWorld4DTraj'), ('pw_ft', 'PreWorld')])
def test_simple_test_on_a_collated_batch_of_two_returns_sample_zero(tag, det):
    """the reference's simple_test indexes batch element 0 of everything it returns (preworld_temporal_traj.py:306, preworld.py:216): fed
    a collated batch of TWO samples, the drop-in must return sample 0's grids -- the reference fixture's, under the near-tie rule --
    while every kernel of the path runs at B = 2 (non-contiguous per-frame pose slices, two-sample lifts, batch-strided decode)."""
    net, dn = _build(det, True, True)
    inputs = tuple(t.to(DEV) for t in E.img_inputs(0, batch=2))
    ego = [[torch.from_numpy(np.concatenate([S.ego_state(40), S.ego_state(47)], 0)).to(DEV)]]

    class TwoSampleDepthNet(type(dn)):                       # sample 0 gets the fixture's seeded DepthNet output, sample 1 another
        def forward(self, x, mlp_input, stereo_metas=None):
            n = x.shape[0] // 2
            a = super().forward(x[:n], mlp_input[:1], stereo_metas)
            self.k -= 1
            self.seed += 5
            b = super().forward(x[n:], mlp_input[1:], stereo_metas)
            self.seed -= 5
            return torch.cat([a, b], 0)
    dn.__class__ = TwoSampleDepthNet
    with torch.no_grad():
        res = net.simple_test(None, None, img=inputs, **(dict(temporal_ego_states=ego) if det == 'PreWorld4DTraj' else {}))
    assert sorted(res.keys()) == list(GOLD[tag + '_keys'])
    for k in res:
        want, got = GOLD[tag + '_' + k], res[k][0]
        assert got.shape == want.shape
        flips = np.nonzero((got != want).reshape(-1))[0]
        if k.startswith('semantic_occ'):
            ti, tm, tc = GOLD['%s_%s_tie_idx' % (tag, k)], GOLD['%s_%s_tie_margin' % (tag, k)], GOLD['%s_%s_tie_cls' % (tag, k)]
            tol = 2 * 2e-5 * float(GOLD['%s_%s_logit_absmax' % (tag, k)])
            for v in flips:
                j = np.nonzero(ti == v)[0]
                assert j.size == 1 and got.reshape(-1)[v] == tc[j[0]] and tm[j[0]] <= tol, (k, int(v))
        else:
            assert flips.size <= 3, (k, flips.size)
