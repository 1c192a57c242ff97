"""CPU: the restatement of the reference's image pipeline (oracle/ref_image_pipeline.py) on cases computed by hand from
OpenCV's published 8-bit resize arithmetic, and the checkpoint dictionary of BasicGANs/utils.py:5-16.  OpenCV itself is
not importable here and the reference holds no image fixtures: parity with cv2 is unpinned (see the oracle's header)."""
import os

import numpy as np
import torch

from oracle import ref_image_pipeline as R


def test_linear_resize_known_answers():
    row = np.array([[[0], [100]]], np.uint8)                        # 1 x 2, one channel
    # x2 up: coordinates -0.25 (clamped to pixel 0), 0.25, 0.75, 1.25 (clamped to pixel 1); weights 1536/512, 512/1536 of 2048
    assert R.resize_linear_u8(row, 4, 1)[0, :, 0].tolist() == [0, 25, 75, 100]
    ramp = np.array([[[10], [20], [30], [40]]], np.uint8)
    assert R.resize_linear_u8(ramp, 2, 1)[0, :, 0].tolist() == [15, 35]         # coordinates 0.5 and 2.5
    col = ramp.transpose(1, 0, 2)                                    # the vertical pass takes the same weights
    assert R.resize_linear_u8(col, 1, 2)[:, 0, 0].tolist() == [15, 35]
    # 3 -> 2: coordinates 0.25 and 1.75: 0.75 * 0 + 0.25 * 90 = 22.5 -> (2048 * (46080 >> 4) >> 16) = 90, (90 + 2) >> 2 = 23
    three = np.array([[[0], [90], [180]]], np.uint8)
    assert R.resize_linear_u8(three, 2, 1)[0, :, 0].tolist() == [23, 158]


def test_resize_to_the_same_size_is_the_identity():
    g = np.random.RandomState(3)
    im = g.randint(0, 256, (13, 9, 3)).astype(np.uint8)
    assert np.array_equal(R.resize_linear_u8(im, 9, 13), im)
    assert np.array_equal(R.resize_nearest_u8(im, 9, 13), im)


def test_nearest_resize_known_answers():
    im = np.arange(6, dtype=np.uint8).reshape(1, 6, 1)
    assert R.resize_nearest_u8(im, 4, 1)[0, :, 0].tolist() == [0, 1, 3, 4]       # floor(d * 1.5)
    assert R.resize_nearest_u8(im, 12, 1)[0, :, 0].tolist() == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5]


def test_prepare_order_of_operations():
    im = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3) * 10      # [H=2, W=3, C=3]
    out = R.prepare(im, (2, 3), (0, 1), (2, 2), flip=True, mean=(0.5, 0.5, 0.5), std=(0.5, 0.25, 1.0))
    window = im[:, 1:3][:, ::-1].transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    want = (window - np.float32(0.5)) / np.array([0.5, 0.25, 1.0], np.float32)[:, None, None]
    assert out.shape == (3, 2, 2) and np.array_equal(out, want)
    mask = R.prepare(im, (4, 6), (1, 2), (2, 3), flip=False, linear=False, divisor=1.0)
    assert np.array_equal(mask, R.resize_nearest_u8(im, 6, 4)[1:3, 2:5].transpose(2, 0, 1).astype(np.float32))


def test_checkpoint_dictionary_round_trip(tmp_path):
    """BasicGANs/utils.py:5-16: one file with net_G / net_D / optimizer_G / optimizer_D; a second one with G's weights."""
    from gan_amd import data
    torch.manual_seed(0)
    make = lambda: (torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3)), torch.nn.Linear(3, 1))
    net_G, net_D = make()
    opt_G, opt_D = torch.optim.Adam(net_G.parameters(), 2e-4, betas=(0.5, 0.999)), torch.optim.Adam(net_D.parameters(), 1e-4)
    net_D(net_G(torch.randn(5, 4))).sum().backward()
    opt_G.step(); opt_D.step()
    path = os.path.join(tmp_path, "ckpt.pth")
    data.save_checkpoint(path, net_G, net_D, opt_G, opt_D)
    raw = torch.load(path, weights_only=True)
    assert sorted(raw) == ["net_D", "net_G", "optimizer_D", "optimizer_G"]
    g2, d2 = make()
    o_g2, o_d2 = torch.optim.Adam(g2.parameters(), 1.0), torch.optim.Adam(d2.parameters(), 1.0)
    data.load_checkpoint(path, g2, d2, o_g2, o_d2)
    for a, b in zip(list(net_G.state_dict().values()) + list(net_D.state_dict().values()),
                    list(g2.state_dict().values()) + list(d2.state_dict().values())):
        assert torch.equal(a, b)
    assert o_g2.param_groups[0]["lr"] == 2e-4 and o_g2.param_groups[0]["betas"] == (0.5, 0.999)
    assert torch.equal(o_g2.state_dict()["state"][0]["exp_avg"], opt_G.state_dict()["state"][0]["exp_avg"])
    wpath = os.path.join(tmp_path, "g.pth")
    data.save_weight(wpath, net_G)
    assert list(torch.load(wpath, weights_only=True)) == list(net_G.state_dict())
