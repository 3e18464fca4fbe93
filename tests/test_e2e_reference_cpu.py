"""The oracle's detector-level composition against tests/golden/e2e_small.npz -- outputs of the REFERENCE'S OWN detector
classes running end to end (tools/gen_golden.py gen_e2e: PreWorld4DTraj.simple_test preworld_temporal_traj.py:212-370,
PreWorld.simple_test preworld.py:159-226, BEVStereo4DOCC.prepare_inputs / extract_img_feat bevdet_occ.py:88-269, imported under
a shim with the image side replaced by the seeded stand-ins of tests/_e2e_stub.py).  Pins what VERDICT r02 listed as
restatement-against-restatement: fp64 pose algebra, frame order, mlp_input from the KEY frame's poses, [adjacent, key] concat
order, with_prev=False zeros, the permutes, the recursion, result keys of both decode branches."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _e2e_stub as E  # noqa: E402
from oracle import oracle as O  # noqa: E402
from preworld_amd import synth as S  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'e2e_small.npz'))


def _inputs():
    return [t.numpy() for t in E.img_inputs(0)]


def test_prepare_inputs_pose_algebra_matches_reference():
    s2k, _, _, _, _, _, c2a = O.prepare_inputs(_inputs())
    np.testing.assert_allclose(np.stack(s2k, 0), GOLD['prep_sensor2keyego'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(np.stack(c2a[:2], 0), GOLD['prep_curr2adjsensor'], rtol=0, atol=2e-6)
    assert c2a[2] is None


@pytest.mark.parametrize('tag,det,post_ft,with_prev', E.RUNS)
def test_oracle_composition_matches_reference_detectors(tag, det, post_ft, with_prev):
    import torch
    dn = E.SeededDepthNet(0)
    mlps = []

    def depthnet(k, mlp):
        mlps.append(mlp)
        assert dn.k == k
        return dn(torch.zeros(E.N_CAMS, 1), torch.from_numpy(mlp)).numpy()
    res, bev, vf = O.detector_simple_test(_inputs(), depthnet, S.ego_state(40), S.synth_state_dict(0), E.GRID, E.INPUT_SIZE, 16,
                                          detector=det, post_finetune=post_ft, with_prev=with_prev,
                                          test_threshold=E.TEST_THRESHOLD)
    assert len(mlps) == int(GOLD[tag + '_n_depthnet_calls'])
    if tag == 'p4d_ft':
        np.testing.assert_allclose(np.stack(mlps, 0), GOLD['mlp_input'], rtol=1e-6, atol=1e-6)
    idx = GOLD['sample_idx']
    bev_rows = bev[0].reshape(32, -1)[:, idx].T
    vf_rows = vf[0].transpose(3, 2, 1, 0).reshape(32, -1)[:, idx].T                 # (B,X,Y,Z,C) -> rows at z*Y*X + y*X + x
    scale = float(np.abs(GOLD[tag + '_bev_rows']).max())
    assert float(np.abs(bev_rows - GOLD[tag + '_bev_rows']).max()) <= 2e-5 * scale
    assert float(np.abs(vf_rows - GOLD[tag + '_vf_rows']).max()) <= 2e-5 * float(np.abs(GOLD[tag + '_vf_rows']).max())
    assert sorted(res.keys()) == list(GOLD[tag + '_keys'])
    for k in res:
        want = GOLD[tag + '_' + k]
        flips = int((res[k][0] != want).sum())
        assert res[k][0].dtype == np.uint8 and res[k][0].shape == want.shape
        assert flips <= 3, (tag, k, flips)            # torch-CPU conv vs the C oracle: fp32 summation order, exact ties only
