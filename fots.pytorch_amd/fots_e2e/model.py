"""FOTSNet -- inference restatement of the reference's `ModelResNetSep2(attention=True)`
(`tools/models.py:237-457`) with the reference's parameter names, so that
`FOTSNet().load_state_dict(reference_net.state_dict())` works key for key.

Layout (all at the strides the reference uses):
  stem      `layer0` (two 3x3 convs, each followed by the concat(x, -x) -> InstanceNorm -> leaky
            "CReLU" of models.py:40-47; stride 2) and `layer0_1` (two 3x3 convs + ReLU; stride 4)
            -> `focr`, the 64-channel 1/4-resolution map RoIRotate samples;
  trunk     `layer1..4`: residual stages of 3/4/6/4 blocks, plain 3x3 convs with InstanceNorm in
            stages 1-2, depthwise-separable convs in stages 3-4 (models.py:139-198);
  merge     1x1 lateral convs `feature1..4`, bilinear upsampling, sigmoid attention gates from one
            shared 1x1 conv `conv_attenton`, depthwise-separable smoothing `upconv1/2` (:405-436);
  heads     `act` (text score), `rbox` (4 distances, x128), `angle` (unit 2-vector) on the 1/4 map
            and on the 1/8 map (:438-457);
  recogniser `forward_ocr` (:334-379): conv5 .. conv11 with two (2,1) max-pools, log-softmax
            over classes, output (N, nclass, T) with T = pooled width.
Only inference is restated (no losses, no weight recomputation hooks).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv(cin, cout, k=3, stride=1, pad=None, groups=1, bias=False):
    if pad is None:
        pad = k // 2 if isinstance(k, int) else tuple(v // 2 for v in k)
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, groups=groups, bias=bias)


def _inorm(ch, affine=True):
    return nn.InstanceNorm2d(ch, eps=1e-05, momentum=0.1, affine=affine)


class _MirrorNorm(nn.Module):
    """concat(x, -x) -> InstanceNorm(2c, affine) -> leaky ReLU  (models.py:40-47, `CReLU_IN`)."""

    def __init__(self, channels):
        super().__init__()
        self.bn = _inorm(2 * channels)

    def forward(self, x):
        return F.leaky_relu(self.bn(torch.cat((x, -x), 1)), 0.01)


class _Residual(nn.Module):
    """out = act(body(x) + shortcut(x)); `body` is supplied by the two block flavours below."""
    slope = 0.0  # ReLU

    def _finish(self, out, x):
        out = out + (x if self.downsample is None else self.downsample(x))
        return F.leaky_relu(out, self.slope) if self.slope else F.relu(out)


class _PlainBlock(_Residual):
    """models.py:139-167 (`BasicBlockIn`): 3x3 conv - IN - ReLU - 3x3 conv - IN."""

    def __init__(self, cin, cout, stride, downsample):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, cout, 3, stride), _inorm(cout)
        self.conv2, self.bn2 = _conv(cout, cout, 3), _inorm(cout)
        self.downsample = downsample

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        return self._finish(self.bn2(self.conv2(out)), x)


class _SeparableBlock(_Residual):
    """models.py:169-198 (`BasicBlockSepIn`): depthwise 3x3 + 1x1 (IN without affine, leaky), then
    depthwise 3x3 - IN - leaky - 1x1 - IN (models.py:85-91, :93-102)."""
    slope = 0.01

    def __init__(self, cin, cout, stride, downsample):
        super().__init__()
        self.conv_sep1 = nn.Sequential(_conv(cin, cin, 3, stride, groups=cin), _conv(cin, cout, 1),
                                       _inorm(cout, affine=False), nn.LeakyReLU(0.01))
        self.conv2 = nn.Sequential(_conv(cout, cout, 3, groups=cout), _inorm(cout), nn.LeakyReLU(0.01),
                                   _conv(cout, cout, 1), _inorm(cout))
        self.downsample = downsample

    def forward(self, x):
        return self._finish(self.conv2(self.conv_sep1(x)), x)


def _stage(block, cin, cout, depth, stride):
    """models.py:318-332: the first block changes stride / width and carries the 1x1 + BatchNorm shortcut."""
    shortcut = None
    if stride != 1 or cin != cout:
        shortcut = nn.Sequential(_conv(cin, cout, 1, stride), nn.BatchNorm2d(cout))
    blocks = [block(cin, cout, stride, shortcut)]
    blocks += [block(cout, cout, 1, None) for _ in range(depth - 1)]
    return nn.Sequential(*blocks)


def _smooth(ch):
    """models.py:70-74 (`conv_dw_plain`)."""
    return nn.Sequential(_conv(ch, ch, 3, groups=ch), _conv(ch, ch, 1))


class FOTSNet(nn.Module):
    def __init__(self, nclass=87, attention=True):
        super().__init__()
        self.attention = attention
        self.layer0 = nn.Sequential(_conv(3, 16), _MirrorNorm(16), _conv(32, 32, 3, 2), _MirrorNorm(32))
        self.layer0_1 = nn.Sequential(_conv(64, 64), nn.ReLU(), _conv(64, 64, 3, 2), nn.ReLU())
        # recogniser
        self.conv5, self.conv6 = _conv(64, 128), _conv(128, 128)
        self.conv7, self.conv8, self.conv9 = _conv(128, 256), _conv(256, 256), _conv(256, 256)
        self.conv10_s = _conv(256, 256, (2, 3), pad=(0, 1))
        self.conv11 = _conv(256, nclass, 1, bias=True)
        # batch6 / batch8 / batch9 hold parameters the reference never applies (models.py:345-364):
        # kept so that checkpoints load key for key
        for name, ch in (("batch5", 128), ("batch6", 128), ("batch7", 256), ("batch8", 256),
                         ("batch9", 256), ("batch10_s", 256)):
            setattr(self, name, _inorm(ch))
        # trunk
        self.layer1 = _stage(_PlainBlock, 64, 64, 3, 1)
        self.layer2 = _stage(_PlainBlock, 64, 128, 4, 2)
        self.layer3 = _stage(_SeparableBlock, 128, 256, 6, 2)
        self.layer4 = _stage(_SeparableBlock, 256, 512, 4, 2)
        # merge
        self.feature4, self.feature3 = _conv(512, 256, 1), _conv(256, 256, 1)
        self.feature2, self.feature1 = _conv(128, 256, 1), _conv(64, 256, 1)
        self.upconv2, self.upconv1 = _smooth(256), _smooth(256)
        # heads
        self.act, self.rbox, self.angle = _conv(256, 1, 1, bias=True), _conv(256, 4, 1, bias=True), _conv(256, 2, 1, bias=True)
        self.drop1 = nn.Dropout2d(p=0.2)
        if attention:
            self.conv_attenton = _conv(256, 1, 1, bias=True)  # (sic) the reference's spelling is the key

    # ------------------------------------------------------------------ shared backbone
    def forward_features(self, x):
        """models.py:381-385: the 64-channel 1/4 map alone."""
        return self.layer0_1(self.layer0(x))

    def _gate(self, t, like):
        """sigmoid(conv_attenton(t)) resized to `like` (models.py:412-416, :423-426, :431-434)."""
        return self._up(torch.sigmoid(self.conv_attenton(t)), like)

    @staticmethod
    def _up(t, like):
        return F.interpolate(t, size=like.shape[2:], mode="bilinear", align_corners=True)

    def _heads(self, t):
        score = torch.sigmoid(self.act(t))
        dist = torch.sigmoid(self.rbox(t)) * 128
        ang = torch.sigmoid(self.angle(t)) * 2 - 1
        ang = ang / torch.sqrt(ang[:, 0] * ang[:, 0] + ang[:, 1] * ang[:, 1]).unsqueeze(1)
        return score, dist, ang

    def forward(self, x):
        """-> [score, score_1/8], [rbox, rbox_1/8], [angle, angle_1/8], [merged 256-ch 1/4 map, focr]
        (models.py:387-457)."""
        focr = self.forward_features(x)
        c1 = self.layer1(self.drop1(focr))
        c2 = self.layer2(c1)
        c3 = self.layer3(c2)
        c4 = self.drop1(self.layer4(c3))
        f1, f2, f3, f4 = self.feature1(c1), self.feature2(c2), self.feature3(c3), self.feature4(c4)

        if self.attention:
            # the first gate is expanded to 256 identical channels before it is resized (:413-415)
            t = self._up(f4, f3) + f3 * self._gate(f4, f3).expand_as(f3)
            gate2 = self._gate(t, f2)
            m2 = self.upconv1(self._up(t, f2)) + f2 * gate2
            gate1 = self._gate(m2, f1)
            m1 = self.upconv2(self._up(m2, f1)) + f1 * gate1
        else:
            t = self._up(f4, f3) + f3
            m2 = self.upconv1(self._up(t, f2)) + f2
            m1 = self.upconv2(self._up(m2, f1)) + f1
        s8, r8, a8 = self._heads(m2)
        m1 = self.drop1(m1)
        s4, r4, a4 = self._heads(m1)
        return [s4, s8], [r4, r8], [a4, a8], [m1, focr]

    # ------------------------------------------------------------------ recogniser
    def forward_ocr(self, x):
        """(N, 64, 11, W) crops -> (N, nclass, W) log-probabilities (models.py:334-379).  conv6, conv8
        and conv9 are each applied twice with shared weights, as the reference does."""
        act = lambda t: F.leaky_relu(t, 0.01)  # noqa: E731
        pool = lambda t: F.max_pool2d(t, (2, 1), stride=(2, 1))  # noqa: E731
        x = act(self.batch5(self.conv5(x)))
        x = act(self.conv6(act(self.conv6(x))))
        x = act(self.batch7(self.conv7(pool(x))))
        x = act(self.conv8(act(self.conv8(x))))
        x = act(self.conv9(act(self.conv9(x))))
        x = act(self.batch10_s(self.conv10_s(pool(x))))
        x = self.conv11(self.drop1(x)).squeeze(2)
        return F.log_softmax(x, dim=1)
