"""Stem convolution (1 -> 18 channels, 7x7x7, stride (2,2,1), pad 3) as MIOpen sees it vs the space-to-depth form:
the 2x2 (y, x) phases of the padded input become 4 input channels, the filter becomes 4x4x7 with stride 1 (taps that fall
outside the 7x7 window are zero).  Same arithmetic; C_in = 4 lets the channels-last kernels use 16-byte loads."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd import miopen_env  # noqa: E402
miopen_env.setup()
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
CL = torch.channels_last_3d


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def s2d_input(x):
    xp = F.pad(x, (3, 3, 3, 3, 3, 3))                                   # [B, 1, 134, 134, 134]
    B, _, Y, X, Z = xp.shape
    return xp.view(B, Y // 2, 2, X // 2, 2, Z).permute(0, 2, 4, 1, 3, 5)  # [B, 2, 2, 67, 67, 134] (p, q) phases


def s2d_weight(w):
    w8 = F.pad(w, (0, 0, 0, 1, 0, 1))                                    # [18, 1, 8, 8, 7]
    O = w8.shape[0]
    return w8.view(O, 4, 2, 4, 2, 7).permute(0, 2, 4, 1, 3, 5).reshape(O, 4, 4, 4, 7)


for B in (8,):
    x = torch.randn(B, 1, 128, 128, 128, device=dev)
    w = (torch.randn(18, 1, 7, 7, 7, device=dev) * 0.05).requires_grad_(True)
    ref = F.conv3d(x, w, None, (2, 2, 1), 3)
    gy = torch.randn_like(ref)

    def alt_fwd(mf):
        xs = s2d_input(x).reshape(B, 4, 67, 67, 134).contiguous(memory_format=mf)
        return F.conv3d(xs, s2d_weight(w).contiguous(memory_format=mf), None, 1, 0)
    for mf, name in ((CL, "channels_last_3d"), (torch.contiguous_format, "contiguous")):
        out = alt_fwd(mf)
        print(name, "max |diff| fwd", (out - ref).abs().max().item(), "of", ref.abs().max().item())
        gw_ref = torch.autograd.grad(ref, w, gy, retain_graph=True)[0]
        gw_alt = torch.autograd.grad(out, w, gy.contiguous(memory_format=mf), retain_graph=True)[0]
        print(name, "max |diff| dW ", (gw_alt - gw_ref).abs().max().item(), "of", gw_ref.abs().max().item())
        xc = x.contiguous(memory_format=mf)
        t_ref_f = timeit(lambda: F.conv3d(xc, w.detach().contiguous(memory_format=mf), None, (2, 2, 1), 3))
        t_alt_f = timeit(lambda: alt_fwd(mf))

        def ref_fb():
            o = F.conv3d(xc, w.contiguous(memory_format=mf), None, (2, 2, 1), 3)
            torch.autograd.grad(o, w, gy)

        def alt_fb():
            torch.autograd.grad(alt_fwd(mf), w, gy)
        t_ref_fb, t_alt_fb = timeit(ref_fb), timeit(alt_fb)
        print("%s  B=%d  stem fwd: MIOpen direct %.0f us, space-to-depth %.0f us | fwd+dW: %.0f us vs %.0f us" % (name, B, t_ref_f, t_alt_f, t_ref_fb, t_alt_fb), flush=True)
