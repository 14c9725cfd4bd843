"""TEST/BENCH INFRASTRUCTURE ONLY -- the ESPNet graph restated over torch's CPU operators.

The reference's CPU path IS torch CPU kernels (nn.Conv2d & co. through MKL-DNN); its Python
source cannot travel to the GPU box, so bench.py's ``cpu_baseline`` times this functional port of
module/espnet/test/Model.py:341-378 on the box's host cores instead (kind "port").  Pinned against
the same golden vectors as the C oracle (tests/test_oracle_golden.py).  Never imported by the
product package.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _bn(x, sd, p):
    return F.batch_norm(x, _t(sd, p + ".running_mean"), _t(sd, p + ".running_var"), _t(sd, p + ".weight"),
                        _t(sd, p + ".bias"), False, 0.0, 1e-3)


def _br(x, sd, p):                                     # Model.py:47-54
    return F.prelu(_bn(x, sd, p + ".bn"), _t(sd, p + ".act.weight"))


def _cbr(x, sd, p, stride=1):                          # Model.py:24-32
    w = _t(sd, p + ".conv.weight")
    return F.prelu(_bn(F.conv2d(x, w, None, stride, (w.shape[2] - 1) // 2), sd, p + ".bn"), _t(sd, p + ".act.weight"))


def _branches(o1, sd, p):                              # Model.py:146-157
    outs = [F.conv2d(o1, _t(sd, "%s.d%d.conv.weight" % (p, d)), None, 1, d, d) for d in (1, 2, 4, 8, 16)]
    add1 = outs[1]
    add2 = add1 + outs[2]
    add3 = add2 + outs[3]
    add4 = add3 + outs[4]
    return torch.cat([outs[0], add1, add2, add3, add4], 1)


def _down(x, sd, p):                                   # Model.py:144-160
    o1 = F.conv2d(x, _t(sd, p + ".c1.conv.weight"), None, 2, 1)
    return F.prelu(_bn(_branches(o1, sd, p), sd, p + ".bn"), _t(sd, p + ".act.weight"))


def _esp(x, sd, p):                                    # Model.py:187-214
    o1 = F.conv2d(x, _t(sd, p + ".c1.conv.weight"))
    return _br(x + _branches(o1, sd, p), sd, p + ".bn")


@torch.no_grad()
def espnet_forward(x, sd, p=2, q=8):
    """x: fp32 [N,3,H,W] CPU tensor -> logits [N,classes,H,W]."""
    e = "encoder."
    out0 = _cbr(x, sd, e + "level1", 2)
    inp1 = F.avg_pool2d(x, 3, 2, 1)
    inp2 = F.avg_pool2d(inp1, 3, 2, 1)
    out0_cat = _br(torch.cat([out0, inp1], 1), sd, e + "b1")
    out1_0 = _down(out0_cat, sd, e + "level2_0")
    out1 = out1_0
    for i in range(p):
        out1 = _esp(out1, sd, e + "level2.%d" % i)
    out1_cat = _br(torch.cat([out1, out1_0, inp2], 1), sd, e + "b2")
    out2_0 = _down(out1_cat, sd, e + "level3_0")
    out2 = out2_0
    for i in range(q):
        out2 = _esp(out2, sd, e + "level3.%d" % i)
    out2_cat = _br(torch.cat([out2_0, out2], 1), sd, e + "b3")
    out2_c = F.conv_transpose2d(_bn(F.conv2d(out2_cat, _t(sd, e + "classifier.conv.weight")), sd, "br"),
                                _t(sd, "up_l3.0.weight"), None, 2)
    out1_c = F.conv2d(out1_cat, _t(sd, "level3_C.conv.weight"))
    t = _cbr(_br(torch.cat([out1_c, out2_c], 1), sd, "combine_l2_l3.0"), sd, "combine_l2_l3.1")
    comb = _br(F.conv_transpose2d(t, _t(sd, "up_l2.0.weight"), None, 2), sd, "up_l2.1")
    feat = _cbr(torch.cat([comb, out0_cat], 1), sd, "conv")
    return F.conv_transpose2d(feat, _t(sd, "classifier.weight"), None, 2)


def preprocess(tiles_u8, mean, std):
    """uint8 [N,H,W,3] BGR -> fp32 [N,3,H,W]; VisualizeResults_iou.py:107-117."""
    x = torch.from_numpy(np.ascontiguousarray(tiles_u8)).to(torch.float32)
    x = (x - torch.tensor(mean, dtype=torch.float32)) / torch.tensor(std, dtype=torch.float32)
    x = x / 255
    return x.permute(0, 3, 1, 2).contiguous()
