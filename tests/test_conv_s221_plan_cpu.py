"""CPU: the WORK PLAN of csrc/conv_s221.hip's weight-gradient kernel, emulated lane by lane in numpy against torch's gradient of the direct
convolution (backbone.py:84: conv(18 -> 18, ks 7, stride (2, 2, 1), pad 3)).  What is emulated is exactly what the kernel does with indices:
the (tap pair, wave) -> (output row, 32-column segment) unit walk, the valid output-column range per kx, the per-lane byte offsets of the A
(input window) and B (output gradient) fragments, the range check of the buffer loads that implements the z padding (negative offsets wrap
to huge unsigned values), the 32x32x2 MFMA fragment / accumulator lane maps, the per-wave partials and the fixed-order finish.  The GPU test
(tests/test_conv_s221_gpu.py) checks the kernel itself; this one pins the arithmetic it was written from, and runs without a GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

G_UNROLL, G_SEG = 8, 32          # csrc/conv_s221.hip


def _buffer_load(base, off_bytes, num_records, arr):
    """raw buffer load, 4 bytes per lane: offsets are unsigned 32-bit; outside [0, num_records) the hardware returns 0"""
    off = off_bytes.astype(np.int64) & 0xFFFFFFFF
    ok = off + 4 <= num_records
    return np.where(ok, arr[np.where(ok, base + off // 4, 0)], 0.0)


def emulate_wgrad(gy, x, B, Y, X, Z, Ci, Co, K, wpp):
    """gy [B, OY, OX, Z, Co], x [B, Y, X, Z, Ci] (channels-last storage) -> grad_weight [Co, K, K, K, Ci]"""
    P, OY, OX = K // 2, Y // 2, X // 2
    nseg = (OX + G_SEG - 1) // G_SEG
    units = B * OY * nseg
    MT = (K * Ci + 31) // 32
    ZCi, ZCo, rows = Z * Ci, Z * Co, K * Ci
    partial = np.zeros((K * K, wpp, rows, Co))
    gyf, xf = gy.reshape(-1), x.reshape(-1)
    lane = np.arange(64)
    c, kk = lane & 31, lane >> 5
    covered = np.zeros((K * K, B * OY * OX), dtype=np.int64)
    for pair in range(K * K):
        ky, kx = divmod(pair, K)
        d = P - kx
        ox_min = (d + 1) // 2 if d > 0 else 0
        ox_max = min((X - 1 + d) // 2, OX - 1)
        tpc = Z // (2 * G_UNROLL)
        offa = [((kk - P) * Ci + m * 32 + c) * 4 for m in range(MT)]
        offb = (kk * Co + np.minimum(c, Co - 1)) * 4
        for wi in range(wpp):
            acc = np.zeros((MT, 32, 32))
            for unit in range(wi, units, wpp):
                row, seg = divmod(unit, nseg)
                b, oy = divmod(row, OY)
                iy = 2 * oy + ky - P
                if iy < 0 or iy >= Y:
                    continue
                lo, hi = max(ox_min, seg * G_SEG), min(ox_max, seg * G_SEG + G_SEG - 1)
                if lo > hi:
                    continue
                xbase = ((b * Y + iy) * X + (2 * lo - d)) * ZCi
                gbase = ((b * OY + oy) * OX + lo) * ZCo
                for col in range(hi - lo + 1):
                    covered[pair, (b * OY + oy) * OX + lo + col] += 1
                    for tr in range(tpc):
                        for u in range(G_UNROLL):
                            bv = _buffer_load(gbase + col * ZCo, offb + (tr * 2 * G_UNROLL + 2 * u) * Co * 4, ZCo * 4, gyf)
                            Bm = np.zeros((2, 32))
                            Bm[kk, c] = bv                       # B fragment: lane <-> (column l & 31, k l >> 5)
                            for m in range(MT):
                                av = _buffer_load(xbase + col * 2 * ZCi, offa[m] + (tr * 2 * G_UNROLL + 2 * u) * Ci * 4, ZCi * 4, xf)
                                Am = np.zeros((32, 2))
                                Am[c, kk] = av                   # A fragment: lane <-> (row l & 31, k l >> 5)
                                acc[m] += Am @ Bm
            for m in range(MT):
                for r in range(16):
                    rowi = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)          # C/D map of the 32x32 MFMA
                    ok = (rowi < rows) & (c < Co)
                    partial[pair, wi][rowi[ok], c[ok]] = acc[m][rowi - m * 32, c][ok]
    gw = np.zeros((Co, K * K, rows))
    for pair in range(K * K):
        gw[:, pair, :] = partial[pair].sum(0).T                   # fixed order over the waves of the pair
    return gw.reshape(Co, K, K, K, Ci), covered


@pytest.mark.parametrize("B,Y,X,Z,Ci,Co,K,wpp", [(1, 4, 8, 16, 18, 18, 7, 4), (2, 4, 6, 32, 5, 3, 3, 8), (1, 2, 68, 16, 4, 2, 3, 4)])
def test_work_plan_of_the_weight_gradient_kernel_equals_autograd(B, Y, X, Z, Ci, Co, K, wpp):
    torch.manual_seed(0)
    x = torch.randn(B, Ci, Y, X, Z, dtype=torch.float64)
    w = torch.randn(Co, Ci, K, K, K, dtype=torch.float64, requires_grad=True)
    y = F.conv3d(x, w, None, (2, 2, 1), K // 2)
    gy = torch.randn_like(y)
    ref, = torch.autograd.grad(y, w, gy)
    got, covered = emulate_wgrad(gy.permute(0, 2, 3, 4, 1).contiguous().numpy(), x.permute(0, 2, 3, 4, 1).contiguous().numpy(), B, Y, X, Z, Ci, Co, K, wpp)
    ref = ref.permute(0, 2, 3, 4, 1).numpy()
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert covered.max() == 1                     # no output column is visited twice by the waves of a tap pair
    P, OY, OX = K // 2, Y // 2, X // 2
    for pair in range(K * K):                     # and exactly the columns whose input column exists are visited
        ky, kx = divmod(pair, K)
        want = sum(1 for b in range(B) for oy in range(OY) for ox in range(OX) if 0 <= 2 * oy + ky - P < Y and 0 <= 2 * ox + kx - P < X)
        assert covered[pair].sum() == want
