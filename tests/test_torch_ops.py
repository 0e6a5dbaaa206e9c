"""`torch.ops.glsdet.*`: the C ABI registered as torch custom ops (glsdet_amd/torch_ops.py).  One test per op,
each called through the dispatcher and held to the oracle; CPU: the schemas exist and a CPU tensor is refused."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import glsdet_oracle as O


def test_ops_are_registered_and_have_no_cpu_fallback():
    import glsdet_amd.torch_ops  # noqa: F401
    for name in ("conv_bn_act", "nonlocal_dot", "yolox_decode", "nms", "batched_nms"):
        assert hasattr(torch.ops.glsdet, name)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        torch.ops.glsdet.batched_nms(torch.zeros(2, 4), torch.zeros(2), torch.zeros(2, dtype=torch.int64), 0.5)
    y = torch.ops.glsdet.conv_bn_act(torch.empty(2, 9, 11, 16, device="meta", dtype=torch.float16), torch.empty(32, 192, device="meta"),
                                     torch.empty(32, device="meta"), torch.empty(32, device="meta"), 20, 3, 3, 2, 1, 1)
    assert tuple(y.shape) == (2, 5, 6, 24)


def _nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 4e-3)])
def test_conv_bn_act_op(dtype, tol):
    from glsdet_amd.torch_ops import pack_conv_weight
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 16, 19, 23, generator=g)
    w = torch.randn(20, 16, 3, 3, generator=g) / 12.0
    scale, bias = torch.rand(20, generator=g) + 0.5, torch.randn(20, generator=g) * 0.3
    r = (lambda t: t.half().float()) if dtype == torch.float16 else (lambda t: t)
    want = O._act(F.conv2d(r(x), r(w), None, 2, 1) * scale[None, :, None, None] + bias[None, :, None, None], "silu")
    res = torch.randn(want.shape, generator=g)
    wp, sc, bi = pack_conv_weight(w, scale, bias, 16, dtype)
    resp = torch.zeros(2, 24, want.shape[2], want.shape[3])
    resp[:, :20] = res
    y = torch.ops.glsdet.conv_bn_act(_nhwc(x, dtype), wp, sc, bi, 20, 3, 3, 2, 1, 1, _nhwc(resp, dtype))
    got = y.float().cpu().permute(0, 3, 1, 2)[:, :20]
    assert tuple(y.shape) == (2, want.shape[2], want.shape[3], 24) and y.dtype == dtype
    assert float((got - (want + r(res))).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    with pytest.raises(RuntimeError):
        torch.ops.glsdet.conv_bn_act(_nhwc(x, dtype)[..., :12], wp, sc, bi, 20, 3, 3, 2, 1, 1)          # not contiguous / C % 8


@pytest.mark.gpu
def test_nonlocal_dot_op(golden):
    from glsdet_amd.torch_ops import pack_conv_weight
    from tests.helpers import block_case
    sd, x, want = block_case(golden, "nonlocal_c16")
    ci = sd["m.theta.weight"].shape[0]
    w = torch.cat([sd["m.theta.weight"], sd["m.phi.weight"], sd["m.g.weight"]], 0)
    b = torch.cat([sd["m.theta.bias"], sd["m.phi.bias"], sd["m.g.bias"]], 0)
    wp, sc, bi = pack_conv_weight(w, torch.ones(3 * ci), b, x.shape[1], torch.float32)
    xs = _nhwc(x, torch.float32)
    tpg = torch.ops.glsdet.conv_bn_act(xs, wp, sc, bi, 3 * ci, 1, 1, 1, 0, 0)
    y = torch.ops.glsdet.nonlocal_dot(xs, tpg, ci, sd["m.conv_out.weight"].reshape(-1, ci).contiguous().cuda(),
                                      sd["m.conv_out.bias"].cuda())
    assert float((y.cpu().permute(0, 3, 1, 2) - want).abs().max()) <= 5e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
def test_yolox_decode_and_nms_ops(golden, shapes):
    from tests.helpers import model_case
    from tests.test_hip_model import _nms_ref
    meta, sd, x, outs, decoded = model_case(golden, shapes, "gl_tiny_seed0")
    H, W = meta["in_shape"][2:]
    levels = []
    for o in outs:
        t = torch.zeros(o.shape[0], o.shape[2], o.shape[3], 16)
        t[..., : o.shape[1]] = o.permute(0, 2, 3, 1)
        levels.append(t.cuda())
    pred = torch.ops.glsdet.yolox_decode(levels, 10, H, W, 0)
    assert float(((pred.cpu() - decoded).abs() / (decoded.abs() + 1.0)).max()) <= 1e-5
    dets, count, status = torch.ops.glsdet.nms(pred, 10, 0, 0.3, 0.5, 1000)
    assert int(status.item()) == 0
    want = _nms_ref(decoded, 10, 0.3, 0.5)
    for i, wd in enumerate(want):
        assert int(count[i]) == len(wd)
        np.testing.assert_array_equal(dets[i, : len(wd), 6].cpu().numpy(), wd[:, 6])
        np.testing.assert_allclose(dets[i, : len(wd), :4].cpu().numpy(), wd[:, :4], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_batched_nms_op_has_torchvisions_contract():
    rng = np.random.default_rng(3)
    n = 700
    c = rng.uniform(50, 400, (n, 2))
    wh = rng.uniform(10, 120, (n, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    boxes[5] = boxes[3]                                                  # an exact duplicate: the lower index survives
    scores = rng.uniform(0, 1, n).astype(np.float32)
    scores[5] = scores[3]
    labels = rng.integers(0, 6, n)
    labels[5] = labels[3]
    keep = torch.ops.glsdet.batched_nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(),
                                        torch.from_numpy(labels).cuda(), 0.5)
    want = O.batched_nms(boxes, scores, labels.astype(np.float32), 0.5)
    assert keep.dtype == torch.int64 and keep.cpu().tolist() == want.tolist()
    assert torch.ops.glsdet.batched_nms(torch.zeros(0, 4).cuda(), torch.zeros(0).cuda(), torch.zeros(0, dtype=torch.int64).cuda(), 0.5).numel() == 0
