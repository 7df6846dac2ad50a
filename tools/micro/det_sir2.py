"""Kernel-level reproducibility of the fused block: two identical calls, outputs compared."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dm_amd_bootstrap  # noqa
from dm_amd.runtime import hip, ops, pack
from test_gpu_sir import make_params, run, nhwc, r16, hash_normal
dev = torch.device("cuda:0")
for (n, c, w) in [(4, 32, 16), (4, 64, 8), (4, 128, 4), (4, 256, 2), (256, 32, 16)]:
    e = 2 * c
    x = r16(hash_normal((n, c, w, w), 31) * 1.2)
    P = make_params(c, e, 5 * c + w)
    gy = nhwc(r16(hash_normal((n, c, w, w), 33)), dev)
    outs = []
    for rep in range(2):
        y, sv = run(dev, x, P, "gelu", (None, None, None), 1.0, record=True)
        d = lambda t: t.to(dev)
        wpd = pack.pack_dgrad(hip.CONV_1X1, P["wp"]).to(dev); wed = pack.pack_dgrad(hip.CONV_1X1, P["we"]).to(dev)
        wdf = P["wd"].reshape(-1, 9).flip(1).t().contiguous().to(dev)
        nwg = ops.sir_workgroups(w * w, n)
        p1, p2, p3 = (torch.zeros((n, 3, ch), device=dev) for ch in (e, e, c))
        dwp = torch.zeros((nwg, 10, e), device=dev)
        dx, d_o, d_e = ops.sir_backward(gy, sv, d(P["g1"]), d(P["b1"]), d(P["g2"]), d(P["b2"]), d(P["g3"]), wpd, wed, wdf, 16, hip.ACT["gelu"], p1, p2, p3, dwp)
        torch.cuda.synchronize()
        outs.append(dict(y=y, **sv, dx=dx, d_o=d_o, d_e=d_e, p1=p1[:, :, :], p2=p2[:, :2], p3=p3, dwp=dwp))
    print((n, c, w), {k: f"{((outs[0][k].float() - outs[1][k].float()).abs().max() / (outs[1][k].float().abs().max() + 1e-20)).item():.1e}" for k in outs[0]})
