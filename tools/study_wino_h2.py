"""Numerical feasibility of a split-fp16 Winograd F(2x2x2, 3x3x3) convolution -- numpy only.
Compares, against a float64 direct convolution of the same inputs, on a 64 -> 64 layer slice (K = 27 * 64 = 1728):
  a) direct form, fp32 multiply-add chain (what the oracle does)
  b) direct form, split-fp16 operands (x = hi + lo, three products, fp32 accumulate)      -- today's k_conv3d_h2
  c) Winograd F(2,3)^3 in fp32                                                            -- today's PW_PRECISION=f32 path
  d) Winograd F(2,3)^3 with the transformed operands split into fp16 hi + lo (transforms in fp32, the transformed input scaled
     by 2^-3 before the split, transformed weights pre-scaled per (point, output channel) by a power of two)
Prints max|err| / max|ref| and the per-element figure q = max |err| / (4e-6 |ref| + 3e-6 rms(ref)) the GPU range tests bound."""
import numpy as np

rs = np.random.RandomState(0)
CIN, COUT, T = 64, 16, 6                       # T^3 tiles of 2x2x2 outputs -> a 12^3 output block, 14^3 input block
D = 2 * T + 2
x = np.maximum(rs.standard_normal((CIN, D, D, D)), 0).astype(np.float32) * 1.7     # post-ReLU activations
w = (rs.standard_normal((COUT, CIN, 3, 3, 3)) * 0.05).astype(np.float32)

Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def direct(xx, ww, dtype):
    out = np.zeros((COUT, 2 * T, 2 * T, 2 * T), dtype)
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                patch = xx[:, kd:kd + 2 * T, kh:kh + 2 * T, kw:kw + 2 * T].astype(dtype)
                out += np.einsum('oc,cdhw->odhw', ww[:, :, kd, kh, kw].astype(dtype), patch).astype(dtype)
    return out


def split16(a):
    hi = a.astype(np.float16)
    lo = (a.astype(np.float64) - hi.astype(np.float64)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def direct_h2(xx, ww):
    amax = np.abs(ww.reshape(COUT, -1)).max(1)
    S = np.exp2(np.floor(np.log2(1023.0 / amax))).astype(np.float32)
    wh, wl = split16(ww * S[:, None, None, None, None])
    xh, xl = split16(xx)
    out = np.zeros((COUT, 2 * T, 2 * T, 2 * T), np.float32)
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                sl = (slice(None), slice(kd, kd + 2 * T), slice(kh, kh + 2 * T), slice(kw, kw + 2 * T))
                for a, b in ((wh, xh), (wl, xh), (wh, xl)):
                    out += np.einsum('oc,cdhw->odhw', a[:, :, kd, kh, kw].astype(np.float64), b[sl].astype(np.float64)).astype(np.float32)
    return out / S[:, None, None, None]


def wino(xx, ww, dtype, split):
    # weights: U = G w G^T along d, h, w (float64 on the host, like pack_conv_weight_wino), then the working type
    U = np.einsum('ad,be,cf,oidef->oiabc', G, G, G, ww.astype(np.float64))
    out = np.zeros((COUT, 2 * T, 2 * T, 2 * T), np.float32)
    Btt, Att = Bt.astype(dtype), At.astype(dtype)
    if split:
        amax = np.abs(U).max(axis=1)                                           # per (o, a, b, c)
        S = np.exp2(np.floor(np.log2(1023.0 / np.maximum(amax, 1e-30))))
        Uh, Ul = split16(U * S[:, None])
    for td in range(T):
        for th in range(T):
            for tw in range(T):
                d = xx[:, 2 * td:2 * td + 4, 2 * th:2 * th + 4, 2 * tw:2 * tw + 4].astype(dtype)
                V = np.einsum('ad,be,cf,idef->iabc', Btt, Btt, Btt, d).astype(dtype)   # adds / subs only: exact op count differs, magnitude x8
                if split:
                    Vh, Vl = split16(V.astype(np.float32) * np.float32(0.125))
                    M = np.zeros((COUT, 4, 4, 4), np.float32)
                    for a, b in ((Uh, Vh), (Ul, Vh), (Uh, Vl)):
                        M += np.einsum('oiabc,iabc->oabc', a.astype(np.float64), b.astype(np.float64)).astype(np.float32)
                    M = M * np.float32(8.0) / S.astype(np.float32)
                else:
                    M = np.einsum('oiabc,iabc->oabc', U.astype(dtype), V).astype(dtype)
                Y = np.einsum('pa,qb,rc,oabc->opqr', Att, Att, Att, M.astype(dtype)).astype(dtype)
                out[:, 2 * td:2 * td + 2, 2 * th:2 * th + 2, 2 * tw:2 * tw + 2] = Y
    return out


ref = direct(x, w, np.float64)
rms = float(np.sqrt((ref ** 2).mean()))
for name, got in (('a) direct fp32', direct(x, w, np.float32)), ('b) direct split-fp16', direct_h2(x, w)),
                  ('c) Winograd fp32', wino(x, w, np.float32, False)), ('d) Winograd split-fp16', wino(x, w, np.float32, True))):
    err = np.abs(got.astype(np.float64) - ref)
    q = float((err / (4e-6 * np.abs(ref) + 3e-6 * rms)).max())
    print('%-24s max|err| / max|ref| %.2e   rms err / rms ref %.2e   q %.2f' % (name, err.max() / np.abs(ref).max(),
                                                                                 np.sqrt((err ** 2).mean()) / rms, q))
