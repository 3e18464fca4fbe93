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
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'e2e_small.npz'))


def _build(det, post_ft, with_prev):
    net = harness.build_model(E.model_cfg(det, post_ft, with_prev), S.synth_state_dict(0), DEV)
    return net, E.install_image_side(net, seed=0)


def test_prepare_inputs_matches_reference():
    net, _ = _build('PreWorld4DTraj', True, True)
    prep = net.prepare_inputs(tuple(t.to(DEV) for t in E.img_inputs(0)), stereo=True)
    np.testing.assert_allclose(torch.stack(prep[1], 0).cpu().numpy(), GOLD['prep_sensor2keyego'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(torch.stack(prep[7][:2], 0).cpu().numpy(), GOLD['prep_curr2adjsensor'], rtol=0, atol=2e-6)
    assert prep[7][2] is None and len(prep[0]) == 3 and tuple(prep[0][0].shape) == (1, E.N_CAMS, 3) + E.INPUT_SIZE


@pytest.mark.parametrize('tag,det,post_ft,with_prev', E.RUNS)
def test_dropin_detectors_match_reference_detectors(tag, det, post_ft, with_prev):
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
        flips = int((got != want).sum())
        print('[e2e] %-14s %-16s flips vs the reference classes: %d / %d' % (tag, k, flips, want.size))
        assert flips <= 3, (tag, k, flips)               # exact ties of the fp32 logits only
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
