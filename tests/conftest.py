import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def load_weights(fold=1):
    z = np.load(os.path.join(GOLDEN, "weights_fold%d.npz" % fold))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def sd1():
    return load_weights(1)


def random_state_dict(p, q, classes=5, seed=0):
    """random-init weights of the ESPNet(classes, p, q) architecture (shapes per Model.py:242-339)"""
    rng = np.random.default_rng(seed)
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = (rng.standard_normal((co, ci, k, k)) * (1.5 / np.sqrt(ci * k * k))).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        sd[name + ".bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        sd[name + ".num_batches_tracked"] = np.array(1, dtype=np.int64)

    def act(name, c):
        sd[name + ".weight"] = rng.uniform(0.05, 0.4, c).astype(np.float32)

    def block(pre, cin, cout, down):
        n = cout // 5
        n1 = cout - 4 * n
        conv(pre + ".c1.conv", n, cin, 3 if down else 1)
        conv(pre + ".d1.conv", n1, n, 3)
        for d in (2, 4, 8, 16):
            conv(pre + ".d%d.conv" % d, n, n, 3)
        if down:
            bn(pre + ".bn", cout)
            act(pre + ".act", cout)
        else:
            bn(pre + ".bn.bn", cout)
            act(pre + ".bn.act", cout)

    e = "encoder."
    conv(e + "level1.conv", 16, 3, 3); bn(e + "level1.bn", 16); act(e + "level1.act", 16)
    bn(e + "b1.bn", 19); act(e + "b1.act", 19)
    block(e + "level2_0", 19, 64, True)
    for i in range(p):
        block(e + "level2.%d" % i, 64, 64, False)
    bn(e + "b2.bn", 131); act(e + "b2.act", 131)
    block(e + "level3_0", 131, 128, True)
    for i in range(q):
        block(e + "level3.%d" % i, 128, 128, False)
    bn(e + "b3.bn", 256); act(e + "b3.act", 256)
    conv(e + "classifier.conv", classes, 256, 1)
    conv("level3_C.conv", classes, 131, 1)
    bn("br", classes)
    conv("conv.conv", classes, 19 + classes, 3); bn("conv.bn", classes); act("conv.act", classes)
    sd["up_l3.0.weight"] = (rng.standard_normal((classes, classes, 2, 2)) * 0.4).astype(np.float32)
    bn("combine_l2_l3.0.bn", 2 * classes); act("combine_l2_l3.0.act", 2 * classes)
    conv("combine_l2_l3.1.conv", classes, 2 * classes, 3); bn("combine_l2_l3.1.bn", classes); act("combine_l2_l3.1.act", classes)
    sd["up_l2.0.weight"] = (rng.standard_normal((classes, classes, 2, 2)) * 0.4).astype(np.float32)
    bn("up_l2.1.bn", classes); act("up_l2.1.act", classes)
    sd["classifier.weight"] = (rng.standard_normal((classes, classes, 2, 2)) * 0.4).astype(np.float32)
    return sd
