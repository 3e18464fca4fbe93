"""CPU tests of the pure-torch pieces of preworld_amd.losses (no HIP calls): the layout-preserving class softmax and the functional
NaN / Inf sanitiser that replaced the reference's in-place masked assignments (mmdet3d/models/detectors/preworld.py:137-138)."""
import numpy as np
import torch

from preworld_amd import losses as L


def _logits(permuted):
    g = torch.Generator().manual_seed(3)
    if permuted:                       # the OccHead's layout on the training path: channels-last memory viewed as (B, C, X, Y, Z)
        return torch.randn(2, 4, 5, 6, 18, generator=g).permute(0, 4, 3, 2, 1)
    return torch.randn(2, 18, 6, 5, 4, generator=g)


def test_softmax_over_classes_in_the_tensors_own_layout():
    for permuted in (True, False):
        x = _logits(permuted).requires_grad_(True)
        y = L._softmax_classes(x)
        ref = torch.softmax(x.detach(), dim=1)
        assert y.shape == ref.shape and torch.allclose(y, ref, rtol=0, atol=1e-7)
        if permuted:
            assert y.stride() == x.stride()                       # no transposing copy: same memory order as the logits
        g = torch.randn(y.shape, generator=torch.Generator().manual_seed(4))
        y.backward(g)
        x2 = x.detach().clone().requires_grad_(True)
        torch.softmax(x2, dim=1).backward(g)
        assert torch.allclose(x.grad, x2.grad, rtol=0, atol=1e-7)


def test_sanitise_equals_the_references_masked_assignments_forward_and_backward():
    x = _logits(True).clone()
    bad = torch.tensor([0, 7, 100, 333, 1000])
    vals = torch.tensor([float('nan'), float('inf'), -float('inf'), float('nan'), float('inf')])
    xs = x.clone()
    idx = np.unravel_index(bad.numpy(), tuple(x.shape))
    xs[idx] = vals
    a = xs.clone().requires_grad_(True)
    ya = L._Sanitise.apply(a)
    # the reference: in place on a copy (clone keeps the graph)
    b = xs.clone().requires_grad_(True)
    yb = b.clone()
    yb[torch.isnan(yb)] = 0
    yb[torch.isinf(yb)] = 0
    assert torch.equal(ya, yb) and ya.stride() == a.stride()
    g = torch.randn(tuple(x.shape), generator=torch.Generator().manual_seed(5))
    ya.backward(g)
    yb.backward(g)
    assert torch.equal(a.grad, b.grad)
    assert float(a.grad[idx].abs().max()) == 0.0 and not torch.isnan(a.grad).any()
    assert torch.isnan(xs[idx][0])                                  # the caller's tensor is left as it was
