"""The entry point that allocates nothing (ddpm_gn_act_forward_static: GroupNorm + activation forward whose many-workgroup
partial sums live in a device array built into the library) against ddpm_gn_act_forward with a caller workspace, through the
C ABI, on two calls in a row, the second with a smaller input (the array keeps the first call's contents).  Bit-exact: both
run the same kernels."""
import ctypes as C

import pytest
import torch

from dm_amd.runtime import hip
from oracle.noise import hash_normal

pytestmark = pytest.mark.gpu


def _desc(x, gamma, beta, res, y, stats, groups, act):
    n, h, w, c = x.shape
    return hip.GnDesc(x=hip.ptr(x), gamma=hip.ptr(gamma), beta=hip.ptr(beta), res=hip.ptr(res), mask=None, mask_scale=1.0, y=hip.ptr(y),
                      stats=hip.ptr(stats), dy=None, part=None, N=n, HW=h * w, C=c, G=groups, act=act, eps=1e-5)


def _both(dev, n, h, w, c, groups, act, seed, with_res):
    x = (hash_normal((n, h, w, c), seed) * 1.5 + 0.2).to(dev).bfloat16()
    res = hash_normal((n, h, w, c), seed + 1).to(dev).bfloat16() if with_res else None
    gamma, beta = (1 + 0.3 * hash_normal((c,), seed + 2)).to(dev), (0.3 * hash_normal((c,), seed + 3)).to(dev)
    outs = []
    for static in (False, True):
        y = torch.full_like(x, float("nan"))
        stats = torch.full((n, groups, 2), float("nan"), device=dev)
        d = _desc(x, gamma, beta, res, y, stats, groups, hip.ACT[act])
        if static:
            # torch's counter only sees torch's caching allocator; the driver's free-memory figure also sees a hipMalloc made
            # inside the library.  (One untimed call first: the very first launch of a kernel may load its code object.)
            hip.call("ddpm_gn_act_forward_static", C.byref(d), hip.stream())
            torch.cuda.synchronize()
            y.fill_(float("nan")); stats.fill_(float("nan"))
            torch.cuda.synchronize()
            before, free_before = torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0]
            hip.call("ddpm_gn_act_forward_static", C.byref(d), hip.stream())
            torch.cuda.synchronize()
            assert torch.cuda.memory_allocated(dev) == before
            assert torch.cuda.mem_get_info(dev)[0] == free_before, "the no-allocation entry point changed the device's free memory"
        else:
            need = hip.lib().ddpm_gn_scratch_floats(C.byref(d), 0)
            assert need > 0, "the test shape must take the many-workgroup path (the one that uses the workspace)"
            ws = torch.empty(need, device=dev)
            d.scratch, d.scratch_floats = ws.data_ptr(), need
            hip.call("ddpm_gn_act_forward", C.byref(d), hip.stream())
        torch.cuda.synchronize()
        outs.append((y, stats))
    (y0, s0), (y1, s1) = outs
    assert torch.isfinite(y1.float()).all() and torch.isfinite(s1).all()
    assert torch.equal(y0, y1) and torch.equal(s0, s1)
    # and against torch fp32 on the same bf16 input, 1.5e-2 of the output scale
    xf = x.float().cpu().permute(0, 3, 1, 2)
    from oracle import ref_cpu as O
    ref = O.act_fn(act)(torch.nn.functional.group_norm(xf, groups, gamma.cpu(), beta.cpu(), eps=1e-5))
    if with_res:
        ref = ref + res.float().cpu().permute(0, 3, 1, 2)
    got = y1.float().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max() <= 1.5e-2 * ref.abs().max()


def test_static_workspace_entry_matches_the_workspace_entry_twice(dev):
    _both(dev, 6, 64, 64, 32, 16, "gelu", 900, True)       # 8 pixel splits per slab
    _both(dev, 3, 48, 48, 32, 16, "silu", 910, False)      # smaller: 5 splits, fewer images; the array still holds the first call's sums


def test_static_workspace_entry_rejects_oversized_shapes(dev):
    n, h, c, groups = 1200, 128, 32, 16                        # 1200 * 16 * 32 splits * 2 floats > 2^20
    d = hip.GnDesc(x=16, gamma=16, beta=16, res=None, mask=None, mask_scale=1.0, y=16, stats=16, dy=None, part=None, N=n, HW=h * h, C=c,
                   G=groups, act=0, eps=1e-5)
    assert hip.lib().ddpm_gn_act_forward_static(C.byref(d), None) != 0
    assert "built-in workspace" in hip.lib().ddpm_last_error().decode()
