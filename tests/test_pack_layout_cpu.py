"""Host-side weight packers against the index formulas include/preworld_hip.h documents (no GPU: the packers are plain torch ops).
A packed 4096-byte tile is four 1024-byte PIECES of 64 lanes x 16 B, so that a wave-wide 16-byte load reads consecutive bytes
(DESIGN.md 4.14)."""
import numpy as np
import torch

from preworld_amd import ops


def test_pack_conv_weight_is_piece_major_and_matches_the_header_formula():
    rs = np.random.RandomState(0)
    cout, cin, k = 40, 64, 3                                    # cout padded to 64 columns
    w = torch.from_numpy(rs.standard_normal((cout, cin, k, k, k)).astype(np.float32))
    wpk = ops.pack_conv_weight(w)
    nch, taps, nt = cin // 32, k ** 3, 2
    assert tuple(wpk.shape) == (nch, taps, nt, 64, 16) and wpk.dtype == torch.float32
    t = wpk.reshape(nch, taps, nt, 4, 64, 4).numpy()             # [ch][tap][nt][q][lane = h*32 + j][e]
    wf = w.reshape(cout, cin, taps).numpy()
    for ch, tap, n_t, q, h, j, e in [(0, 0, 0, 0, 0, 0, 0), (1, 13, 0, 3, 1, 31, 3), (0, 26, 1, 2, 0, 7, 1), (1, 5, 1, 1, 1, 7, 2)]:
        n, c = n_t * 32 + j, ch * 32 + h * 16 + 4 * q + e
        want = wf[n, c, tap] if n < cout else 0.0
        assert t[ch, tap, n_t, q, h * 32 + j, e] == want, (ch, tap, n_t, q, h, j, e)
    assert np.all(t[:, :, 1, :, [8 + 32 * hh for hh in (0, 1)], :] == 0)      # column 40 (nt 1, j 8) is padding
    # a piece is 1024 consecutive bytes: 64 lanes x 4 floats
    assert wpk.is_contiguous() and wpk[0, 0, 0].reshape(4, 256).stride() == (256, 1)


def test_pack_conv_weight_h2_pieces_planes_and_prescale():
    rs = np.random.RandomState(1)
    cout, cin = 32, 32
    w = torch.from_numpy((rs.standard_normal((cout, cin, 3, 3, 3)) * np.exp(rs.uniform(-6, 2, (cout, 1, 1, 1, 1)))).astype(np.float32))
    wpk, inv = ops.pack_conv_weight_h2(w)
    assert tuple(wpk.shape) == (1, 27, 1, 64, 16) and tuple(inv.shape) == (32,)
    halves = wpk.view(torch.float16).reshape(1, 27, 1, 4, 64, 8).float().numpy()   # [ch][tap][nt][q = 2 ks + p][lane][e]
    S = 1.0 / inv.double().numpy()
    assert np.all(np.log2(S) == np.round(np.log2(S)))                               # powers of two
    amax = np.abs(w.numpy().reshape(cout, -1)).max(1) * S
    assert np.all((amax >= 512) & (amax < 1024))
    wf = w.double().numpy().reshape(cout, cin, 27)
    worst = 0.0
    for tap in (0, 13, 26):
        for ks in (0, 1):
            for h in (0, 1):
                for j in (0, 5, 31):
                    hi = halves[0, tap, 0, 2 * ks + 0, h * 32 + j].astype(np.float64)
                    lo = halves[0, tap, 0, 2 * ks + 1, h * 32 + j].astype(np.float64)
                    want = wf[j, 16 * ks + 8 * h:16 * ks + 8 * h + 8, tap] * S[j]
                    assert np.all(hi == want.astype(np.float16).astype(np.float64))  # plane 0 = fp16(S w)
                    worst = max(worst, float(np.abs(hi + lo - want).max() / max(np.abs(want).max(), 1e-30)))
    assert worst <= 2.0 ** -21, worst                                               # hi + lo carries 22 bits of S w
