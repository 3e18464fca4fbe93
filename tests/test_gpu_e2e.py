"""Row H end to end on the GPU: restated config -> state dict by reference key names -> lifted
inputs -> 7 states -> stacked {0,2,4,6} -> temporal mIoU, against the CPU oracle pipeline on a
fixed synthetic mini-split (C1-sized grid: 1 camera, 100x100x8)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from preworld_amd import harness, synth as S
from _parity import check_argmax

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GC = S.GRID_CONFIG_C1


def _oracle_sample(seed, sd):
    bevs = []
    for f in range(2):
        depth, feat = S.lift_inputs(seed * 16 + f, N=1)
        r = S.synthetic_rig(1, dx=-2.5 * f)
        bev = O.lss_view_transform(depth, feat, r['sensor2ego'], r['intrin'], r['post_rot'], r['post_tran'],
                                   r['bda'], GC, S.INPUT_SIZE, S.DOWNSAMPLE)
        bevs.append(O.pre_process(bev, sd))
    x = O.encoder_forward(bevs[1], bevs[0], sd)            # cat([adjacent, key])
    vf = O.final_conv(x, sd)
    states, feats = O.preworld4d_decode(vf, S.ego_state(seed), sd, n_steps=6, post_finetune=True)
    return states, [O.occ_decode(f, sd)[1] for f in feats]


LOGIT_TIE = 2e-4          # a flipped voxel must have an oracle top-2 margin below this (measured: <= 8e-6 on <= 1 flip per state)


def test_mini_split_states_and_miou_match_oracle():
    sd = S.synth_state_dict(0)
    net = harness.build_model(harness.model_cfg(GC), sd, DEV)
    rs = np.random.RandomState(77)
    samples, oracle_states, oracle_logits = [], [], []
    for seed in (1, 2):
        gt = {h: rs.randint(0, 18, size=(100, 100, 8)).astype(np.uint8) for h in (0, 2, 4, 6)}
        mask = rs.rand(100, 100, 8) < 0.7
        samples.append(dict(frames=harness.lifted_frames(seed, 1, DEV), ego=torch.from_numpy(S.ego_state(seed)).to(DEV),
                            gt=gt, mask_camera=mask))
        st_, lg_ = _oracle_sample(seed, sd)
        oracle_states.append(st_)
        oracle_logits.append(lg_)
    miou, stacks, metric = harness.evaluate(net, samples, DEV)

    # states: argmax agreement per stacked horizon
    for i, (st, ost) in enumerate(zip(stacks, oracle_states)):
        assert st.shape == (4, 100, 100, 8) and st.dtype == np.uint8
        for j, h in enumerate((0, 2, 4, 6)):
            check_argmax('e2e C1 sample %d state %ds' % (i, h), st[j], ost[h], oracle_logits[i][h], LOGIT_TIE)

    # temporal mIoU restated with the oracle's metric on the oracle's states
    want = {}
    for h in (0, 2, 4, 6):
        m = O.MetricMIoU(num_classes=18, use_image_mask=True)
        for s, ost in zip(samples, oracle_states):
            stack = np.stack([ost[k] for k in (0, 2, 4, 6)])
            m.add_batch(stack[h // 2], s['gt'][h], None, s['mask_camera'])
        want[h] = m.count_miou()[0]
    for h in (0, 2, 4, 6):
        assert abs(miou[h] - want[h]) <= 0.05, (h, miou[h], want[h])
    assert abs(miou['avg_future'] - round(float(np.mean([want[2], want[4], want[6]])), 2)) <= 0.05
    # the reference's own return values (occ_metrics.py:548-575): (per-class IoU at 1 s, [mIoU 1 s, 2 s, 3 s])
    iu1, lst = metric.count_miou()
    assert lst == [miou[2], miou[4], miou[6]] and iu1.shape == (18,) and metric.cnt == 2
    assert len(metric.count_iou()) == 3


def _oracle_sample_full(seed, sd):
    """one FULL-SIZE C3 sample (6 cameras, 200x200x16, key + adjacent frame) through the oracle: states 0/2/4/6 + their logits"""
    gc = S.GRID_CONFIG_FULL
    bevs = []
    for f in range(2):
        depth, feat = S.lift_inputs(seed * 16 + f, N=6)
        r = S.synthetic_rig(6, dx=-2.5 * f)
        bev = O.lss_view_transform(depth, feat, r['sensor2ego'], r['intrin'], r['post_rot'], r['post_tran'], r['bda'], gc,
                                   S.INPUT_SIZE, S.DOWNSAMPLE)
        bevs.append(O.pre_process(bev, sd))
    vf = O.final_conv(O.encoder_forward(bevs[1], bevs[0], sd), sd)
    states, feats = O.preworld4d_decode(vf, S.ego_state(seed), sd, n_steps=6, post_finetune=True)
    return {h: states[h] for h in (0, 2, 4, 6)}, {h: O.occ_decode(feats[h], sd)[1] for h in (0, 2, 4, 6)}


def test_fullsize_mini_split_miou_reproduces_the_oracle():
    """north_star: "mIoU on a fixed mini-split is reproduced" -- a split that can FAIL: 8 full-size C3
    samples (6 cameras, 200x200x16, 7 states).  Ground truth = the occupancy a PERTURBED copy of the weights predicts for the
    same inputs, so the model under test scores a two-digit mIoU that depends on its predictions (a wrong model lands
    somewhere else entirely; random labels would give ~1/18 whatever the model does).  The drop-in's temporal mIoU per
    horizon (occ_metrics.py:413-594 through harness.evaluate) must equal the oracle pipeline's to 0.01, and every voxel the
    two disagree on must be a near-tie of the oracle's logits."""
    gc = S.GRID_CONFIG_FULL
    sd = S.synth_state_dict(0)
    net = harness.build_model(harness.model_cfg(gc), sd, DEV)
    rs = np.random.RandomState(123)
    sd_gt = dict(sd)
    for k in sd:
        if k.startswith(('occupancy_head.occ_pred_conv', 'fusion_head.2', 'final_conv.conv.weight')) and k.endswith('weight'):
            sd_gt[k] = (sd[k] + 0.06 * float(np.abs(sd[k]).mean()) * rs.standard_normal(sd[k].shape)).astype(np.float32)
    net_gt = harness.build_model(harness.model_cfg(gc), sd_gt, DEV)
    seeds = list(range(1, 9))
    samples = []
    for seed in seeds:
        frames = harness.lifted_frames(seed, 6, DEV)
        ego = torch.from_numpy(S.ego_state(seed)).to(DEV)
        with torch.no_grad():
            lab = net_gt.simple_test_from_lift(frames, ego, n_steps=6)
        gt = {h: lab['semantic_occ_%ds' % h][0].cpu().numpy() for h in (0, 2, 4, 6)}
        samples.append(dict(frames=frames, ego=ego, gt=gt, mask_camera=rs.rand(200, 200, 16) < 0.7))
    del net_gt
    miou, stacks, metric = harness.evaluate(net, samples, DEV)
    assert metric.cnt == len(seeds)
    want = {h: O.MetricMIoU(num_classes=18, use_image_mask=True) for h in (0, 2, 4, 6)}
    flips = 0
    for i, (seed, s) in enumerate(zip(seeds, samples)):
        ost, olg = _oracle_sample_full(seed, sd)
        for j, h in enumerate((0, 2, 4, 6)):
            check_argmax('mini-split sample %d state %ds' % (seed, h), stacks[i][j], ost[h], olg[h], LOGIT_TIE, floor=0.99999)
            flips += int((stacks[i][j] != ost[h]).sum())
            want[h].add_batch(ost[h], s['gt'][h], None, s['mask_camera'])
    got = {h: miou[h] for h in (0, 2, 4, 6)}
    ref = {h: want[h].count_miou()[0] for h in (0, 2, 4, 6)}
    print('[mini-split] 8 full-size samples: temporal mIoU drop-in %s, oracle %s, %d of %d voxels differ'
          % (got, ref, flips, 8 * 4 * 640000))
    for h in (0, 2, 4, 6):
        assert 10.0 <= ref[h] <= 98.0, ('the split must discriminate', h, ref[h])      # (a 25 % weight perturbation: 12.8 - 19.8)
        assert abs(got[h] - ref[h]) <= 0.01, (h, got[h], ref[h])
    assert abs(miou['avg_future'] - round(float(np.mean([ref[2], ref[4], ref[6]])), 2)) <= 0.01


def test_geo_occ_comes_from_the_same_kernel():
    """preworld_temporal_traj.py:313-319: geo_occ = 0 where the argmax is not the empty class, 17 elsewhere."""
    sd = S.synth_state_dict(0)
    net = harness.build_model(harness.model_cfg(GC), sd, DEV)
    with torch.no_grad():
        res = net.simple_test_from_lift(harness.lifted_frames(3, 1, DEV), torch.from_numpy(S.ego_state(3)).to(DEV), n_steps=2)
    for k in range(3):
        occ = res['semantic_occ_%ds' % k][0].cpu().numpy()
        geo = res['geo_occ_%ds' % k][0].cpu().numpy()
        assert geo.dtype == np.uint8 and geo.shape == occ.shape
        np.testing.assert_array_equal(geo, np.where(occ != 17, 0, 17).astype(np.uint8))


def test_captured_sample_replays_like_eager():
    """pipeline.CapturedSample: a hipGraph replay with new inputs copied into the static buffers gives exactly the results of
    the same pass launched eagerly under the runner's activation-range table (all kernels are deterministic; `cap.eager()`),
    and the detector's own eager entry point -- which calibrates its own table, so a tensor's exponent may differ by a few
    units -- agrees up to a handful of exact ties."""
    from preworld_amd.pipeline import CapturedSample
    sd = S.synth_state_dict(0)
    net = harness.build_model(harness.model_cfg(GC), sd, DEV)
    f1, e1 = harness.lifted_frames(1, 1, DEV), torch.from_numpy(S.ego_state(1)).to(DEV)
    f2, e2 = harness.lifted_frames(2, 1, DEV), torch.from_numpy(S.ego_state(2)).to(DEV)
    cap = CapturedSample(net, f1, e1, n_steps=6)

    def grids(res):
        return {k: v[0].clone() for k, v in res.items() if k.startswith('semantic_occ')}
    with torch.no_grad():
        own2 = grids(net.simple_test_from_lift(f2, e2, n_steps=6))
        own1 = grids(net.simple_test_from_lift(f1, e1, n_steps=6))
    got2 = grids(cap.run(f2, e2))
    want2 = grids(cap.eager())
    got1 = grids(cap.run(f1, e1))
    want1 = grids(cap.eager())
    torch.cuda.synchronize()
    assert cap.ranges_ok()
    assert len(got1) == 7
    for k in want1:
        assert torch.equal(got1[k], want1[k]), k
        assert torch.equal(got2[k], want2[k]), k
        assert int((got1[k] != own1[k]).sum()) <= 8 and int((got2[k] != own2[k]).sum()) <= 8, k
    assert any(not torch.equal(got1[k], got2[k]) for k in got1)
    # the host payload (d2h=True): the OccHead kernels write the 14 (X,Y,Z)-contiguous grids in place into one buffer (no gather copy
    # in the step); its rows equal the result dict's grids, in key order
    capd = CapturedSample(net, f1, e1, n_steps=6, d2h=True)
    res = capd.run(f2, e2)
    torch.cuda.synchronize()
    assert 'grids' in res and len(capd.host_keys) == 14 and tuple(capd.host.shape) == (14,) + tuple(res['semantic_occ_0s'][0].shape)
    for i, k in enumerate(capd.host_keys):
        assert res[k][0].is_contiguous() and torch.equal(capd.host[i], res[k][0].cpu()), k
        if k.startswith('semantic_occ'):
            assert torch.equal(res[k][0], got2[k]), k


def test_two_samples_in_flight_do_not_interfere():
    """bench.py's throughput mode: two captured samples replayed concurrently on two HIP streams must
    produce what each produces alone (no shared scratch between the graphs)."""
    from preworld_amd.pipeline import CapturedSample
    sd = S.synth_state_dict(0)
    net = harness.build_model(harness.model_cfg(GC), sd, DEV)
    caps = [CapturedSample(net, harness.lifted_frames(s_, 1, DEV), torch.from_numpy(S.ego_state(s_)).to(DEV)) for s_ in (1, 2)]
    alone = []
    for c in caps:
        c.replay()
        torch.cuda.synchronize()
        alone.append({k: v[0].clone() for k, v in c.out.items() if k.startswith('semantic_occ')})
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(5):
        for c, st in zip(caps, streams):
            with torch.cuda.stream(st):
                c.replay()
    torch.cuda.synchronize()
    for c, want in zip(caps, alone):
        for k, v in want.items():
            assert torch.equal(c.out[k][0], v), k


def test_pretrain_attribute_decode_branch_matches_oracle():
    """if_post_finetune=False (preworld_temporal_traj.py:224-301): density-threshold + semantic-MLP decode of
    the 7 states, named 0s, 2s..7s like the reference, against the oracle's attribute decode.  The synthetic
    density MLP is biased so that both sides of the 8.5 threshold occur."""
    sd = S.synth_state_dict(0)
    sd['density_mlp.2.bias'] = sd['density_mlp.2.bias'] + np.float32(8.5)
    net = harness.build_model(harness.model_cfg(GC, if_post_finetune=False), sd, DEV)
    frames = harness.lifted_frames(4, 1, DEV)
    with torch.no_grad():
        res = net.simple_test_from_lift(frames, torch.from_numpy(S.ego_state(4)).to(DEV), n_steps=6)
    bevs = []
    for f in range(2):
        depth, feat = S.lift_inputs(4 * 16 + f, N=1)
        r = S.synthetic_rig(1, dx=-2.5 * f)
        bev = O.lss_view_transform(depth, feat, r['sensor2ego'], r['intrin'], r['post_rot'], r['post_tran'],
                                   r['bda'], GC, S.INPUT_SIZE, S.DOWNSAMPLE)
        bevs.append(O.pre_process(bev, sd))
    vf = O.final_conv(O.encoder_forward(bevs[1], bevs[0], sd), sd)
    states, feats = O.preworld4d_decode(vf, S.ego_state(4), sd, n_steps=6, post_finetune=False)
    names = [0, 2, 3, 4, 5, 6, 7]
    assert sorted(k for k in res if k.startswith('semantic_occ')) == sorted('semantic_occ_%ds' % n for n in names)
    for k, n in enumerate(names):
        got = res['semantic_occ_%ds' % n][0].cpu().numpy()
        assert got.shape == states[k].shape
        # a flip needs the density to sit at the 8.5 threshold or the semantic argmax to be a near-tie
        od, osem = O.attribute_decode(feats[k], sd)[1:]
        osem_s = np.sort(osem, -1)
        tie = np.minimum(np.abs(od - 8.5)[0], (osem_s[..., -1] - osem_s[..., -2])[0])
        flips = got != states[k]
        print('[parity] attribute decode state %ds: agreement %.6f, largest tie distance among %d flips %.3e'
              % (n, 1 - flips.mean(), int(flips.sum()), float(tie[flips].max()) if flips.any() else 0.0))
        assert 1 - flips.mean() >= 0.9999 and (not flips.any() or float(tie[flips].max()) <= 2e-3)
    occ0 = res['semantic_occ_0s'][0]
    assert 0.02 < float((occ0 != 17).float().mean()) < 0.98       # both branches of the threshold are exercised


def test_build_model_rejects_incomplete_state_dict():
    sd = S.synth_state_dict(0)
    sd.pop('final_conv.conv.weight')
    with pytest.raises(KeyError):
        harness.build_model(harness.model_cfg(GC), sd, DEV)
