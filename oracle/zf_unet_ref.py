"""CPU restatement of ZF_UNET (lib/models/zf_unet.py:5-95).  TEST INFRASTRUCTURE.

Functional form over a flat ``state_dict`` (name -> tensor) so that golden
fixtures (which ARE state_dicts) drive it directly and so that it shares no
structure with the product's nn.Module shells.

Topology restated from the reference:
  * block(name, cin, cout) = [conv3x3 p1 + bias -> BatchNorm2d(eps 1e-5, momentum 0.1) -> ReLU] x 2
    then Dropout2d (zf_unet.py:20-32; conv :8, bn :9, relu :10)
  * encoder widths f*(1,2,4,8,16,32) at strides 1..32, MaxPool2d(2) between (zf_unet.py:41,44-50,61-76)
  * decoder: cat([nearest-x2(prev), skip], C) -> block (zf_unet.py:42,52-56,78-91)
  * 1x1 head (zf_unet.py:58,93)

Dropout2d is replayed from explicit per-(n, c) multiplier tables (``drop``:
block name -> float tensor [N, C] holding 0 or 1/(1-p)); ``None`` = identity.
"""
import torch
import torch.nn.functional as F

ENCODER = ['conv_224', 'conv_112', 'conv_56', 'conv_28', 'conv_14', 'conv_7']
DECODER = ['up_conv_14', 'up_conv_28', 'up_conv_56', 'up_conv_112', 'up_conv_224']
BLOCKS = ENCODER + DECODER
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def block_channels(filters=32, input_channels=3):
    """(cin, cout) of every double-conv block, in BLOCKS order."""
    f = filters
    enc = [(input_channels, f), (f, 2 * f), (2 * f, 4 * f), (4 * f, 8 * f), (8 * f, 16 * f), (16 * f, 32 * f)]
    dec = [(48 * f, 16 * f), (24 * f, 8 * f), (12 * f, 4 * f), (6 * f, 2 * f), (3 * f, f)]
    return dict(zip(BLOCKS, enc + dec))


def state_shapes(filters=32, input_channels=3, num_classes=1, batch_norm=True):
    """Ordered {name: shape} of the reference state_dict (probe: 156 entries at defaults)."""
    out = {}
    for blk, (cin, cout) in block_channels(filters, input_channels).items():
        for l, ci in (('l1', cin), ('l2', cout)):
            p = '%s.%s.' % (blk, l)
            out[p + 'conv.weight'] = (cout, ci, 3, 3)
            out[p + 'conv.bias'] = (cout,)
            if batch_norm:
                out[p + 'bn.weight'] = (cout,)
                out[p + 'bn.bias'] = (cout,)
                out[p + 'bn.running_mean'] = (cout,)
                out[p + 'bn.running_var'] = (cout,)
                out[p + 'bn.num_batches_tracked'] = ()
    out['conv_final.weight'] = (num_classes, filters, 1, 1)
    out['conv_final.bias'] = (num_classes,)
    return out


def is_param(name):
    return not (name.endswith('running_mean') or name.endswith('running_var')
                or name.endswith('num_batches_tracked'))


def closed_form_fill(sd, seed=0.0):
    """Deterministic, framework-independent fill of a state_dict.

    conv weights  : a*sin(0.37*i + c) with a = sqrt(2/fan_in) so activations stay O(1)
    conv biases   : 0.05*sin(0.91*i + c)
    bn weight     : 1 + 0.1*sin(0.53*i + c);  bn bias : 0.1*cos(0.71*i + c)
    running_mean  : 0;  running_var : 1;  num_batches_tracked : 0
    where i is the flat element index and c = seed + ordinal of the tensor.
    """
    for k, (name, t) in enumerate(sd.items()):
        c = float(seed) + float(k)
        n = t.numel()
        i = torch.arange(n, dtype=torch.float64)
        if name.endswith('conv.weight') or name == 'conv_final.weight':
            fan_in = t.shape[1] * t.shape[2] * t.shape[3]
            v = (2.0 / fan_in) ** 0.5 * 1.7 * torch.sin(0.37 * i + c)
        elif name.endswith('conv.bias') or name == 'conv_final.bias':
            v = 0.05 * torch.sin(0.91 * i + c)
        elif name.endswith('bn.weight'):
            v = 1 + 0.1 * torch.sin(0.53 * i + c)
        elif name.endswith('bn.bias'):
            v = 0.1 * torch.cos(0.71 * i + c)
        elif name.endswith('running_mean'):
            v = torch.zeros(n, dtype=torch.float64)
        elif name.endswith('running_var'):
            v = torch.ones(n, dtype=torch.float64)
        elif name.endswith('num_batches_tracked'):
            v = torch.zeros(n, dtype=torch.float64)
        else:
            raise KeyError(name)
        with torch.no_grad():
            t.copy_(v.reshape(t.shape).to(t.dtype))
    return sd


def default_init_state(filters=32, input_channels=3, num_classes=1, batch_norm=True, seed=0):
    """State dict with torch's DEFAULT initialisation, bit-identical to ``torch.manual_seed(seed);
    ZF_UNET(filters=...)`` of the reference: the reference creates, in this order, two
    nn.Conv2d(cin, cout, 3, padding=1) per block (zf_unet.py:8,23-24; blocks in the order of :44-56) and
    the 1x1 head (:58); BatchNorm2d / Dropout2d / pooling draw nothing from the RNG.
    tests/golden/make_golden.py asserts the equality against the imported reference."""
    sd = {}
    torch.manual_seed(seed)
    for blk, (cin, cout) in block_channels(filters, input_channels).items():
        for l, ci in (('l1', cin), ('l2', cout)):
            conv = torch.nn.Conv2d(ci, cout, 3, padding=1)
            p = '%s.%s.' % (blk, l)
            sd[p + 'conv.weight'] = conv.weight.detach().clone()
            sd[p + 'conv.bias'] = conv.bias.detach().clone()
            if batch_norm:
                sd[p + 'bn.weight'] = torch.ones(cout)
                sd[p + 'bn.bias'] = torch.zeros(cout)
                sd[p + 'bn.running_mean'] = torch.zeros(cout)
                sd[p + 'bn.running_var'] = torch.ones(cout)
                sd[p + 'bn.num_batches_tracked'] = torch.zeros((), dtype=torch.int64)
    head = torch.nn.Conv2d(filters, num_classes, 1)
    sd['conv_final.weight'] = head.weight.detach().clone()
    sd['conv_final.bias'] = head.bias.detach().clone()
    return sd


def new_state(filters=32, input_channels=3, num_classes=1, batch_norm=True, seed=0.0):
    sd = {}
    for name, shape in state_shapes(filters, input_channels, num_classes, batch_norm).items():
        dt = torch.int64 if name.endswith('num_batches_tracked') else torch.float32
        sd[name] = torch.zeros(shape, dtype=dt)
    return closed_form_fill(sd, seed)


def _conv_bn_relu(sd, p, x, train):
    y = F.conv2d(x, sd[p + 'conv.weight'], sd[p + 'conv.bias'], padding=1)
    if (p + 'bn.weight') in sd:
        y = F.batch_norm(y, sd[p + 'bn.running_mean'], sd[p + 'bn.running_var'],
                         sd[p + 'bn.weight'], sd[p + 'bn.bias'],
                         training=train, momentum=BN_MOMENTUM, eps=BN_EPS)
        if train:
            sd[p + 'bn.num_batches_tracked'] += 1
    return torch.relu(y)


def _block(sd, name, x, train, drop):
    x = _conv_bn_relu(sd, name + '.l1.', x, train)
    x = _conv_bn_relu(sd, name + '.l2.', x, train)
    m = None if drop is None else drop.get(name)
    if m is not None:
        x = x * m[:, :, None, None]
    return x


def forward(sd, x, train=False, drop=None):
    """logits = ZF_UNET(x).  In train mode BN running stats in ``sd`` are updated in place."""
    skips = []
    h = x
    for i, name in enumerate(ENCODER):
        h = _block(sd, name, h, train, drop)
        if i + 1 < len(ENCODER):
            skips.append(h)
            h = F.max_pool2d(h, 2)
    for name in DECODER:
        up = F.interpolate(h, scale_factor=2, mode='nearest')
        h = _block(sd, name, torch.cat([up, skips.pop()], dim=1), train, drop)
    return F.conv2d(h, sd['conv_final.weight'], sd['conv_final.bias'])


def make_dropout_tables(filters, batch, p, generator):
    """Per-block [N, C] multiplier tables (0 or 1/(1-p)) -- the Dropout2d replay format."""
    out = {}
    for name, (_, cout) in block_channels(filters).items():
        keep = (torch.rand(batch, cout, generator=generator) >= p).to(torch.float32)
        out[name] = keep / (1.0 - p)
    return out
