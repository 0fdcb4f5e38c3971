"""CPU restatement of FCDenseNet (lib/models/tiramisu.py:9-184).  TEST INFRASTRUCTURE.

Functional form over a state_dict with the reference's key names; pinned against a fixture produced by the
imported reference (tests/golden/make_golden.py gen_tiramisu; tests/test_oracle_golden.py).

  DenseLayer      BN -> ReLU -> conv3x3(growth) -> Dropout2d            tiramisu.py:9-19
  DenseBlock      x = cat([x, layer(x)]); upsample blocks return only the new features   :22-44
  TransitionDown  BN -> ReLU -> conv1x1 -> Dropout2d -> MaxPool2d(2)     :47-59
  TransitionUp    ConvTranspose2d(3, stride 2, pad 0) -> center_crop -> cat([out, skip])  :62-90
Dropout2d is taken as identity (p = 0 on both sides in the fixture; the replay mechanism is pinned on ZF_UNET).
"""
import torch
import torch.nn.functional as F


def _bn(sd, p, x, train):
    y = F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'], sd[p + 'bias'],
                     training=train, momentum=0.1, eps=1e-5)
    if train:
        sd[p + 'num_batches_tracked'] += 1
    return y


def _dense_block(sd, prefix, n_layers, x, train, upsample):
    new = []
    for l in range(n_layers):
        p = '%s.layers.%d.' % (prefix, l)
        out = F.conv2d(torch.relu(_bn(sd, p + 'norm.', x, train)), sd[p + 'conv.weight'], sd[p + 'conv.bias'], padding=1)
        x = torch.cat([x, out], 1)
        new.append(out)
    return torch.cat(new, 1) if upsample else x


def forward(sd, x, down_blocks, up_blocks, bottleneck_layers, train=True):
    out = F.conv2d(x, sd['firstconv.weight'], sd['firstconv.bias'], padding=1)
    skips = []
    for i, n in enumerate(down_blocks):
        out = _dense_block(sd, 'denseBlocksDown.%d' % i, n, out, train, False)
        skips.append(out)
        p = 'transDownBlocks.%d.' % i
        out = F.conv2d(torch.relu(_bn(sd, p + 'norm.', out, train)), sd[p + 'conv.weight'], sd[p + 'conv.bias'])
        out = F.max_pool2d(out, 2)
    out = _dense_block(sd, 'bottleneck.bottleneck', bottleneck_layers, out, train, True)
    for i, n in enumerate(up_blocks):
        skip = skips.pop()
        p = 'transUpBlocks.%d.convTrans.' % i
        up = F.conv_transpose2d(out, sd[p + 'weight'], sd[p + 'bias'], stride=2)
        h, w = skip.shape[2], skip.shape[3]
        y0, x0 = (up.shape[2] - h) // 2, (up.shape[3] - w) // 2
        up = up[:, :, y0:y0 + h, x0:x0 + w]
        out = _dense_block(sd, 'denseBlocksUp.%d' % i, n, torch.cat([up, skip], 1), train, i + 1 < len(up_blocks))
    return F.conv2d(out, sd['finalConv.weight'], sd['finalConv.bias'])
