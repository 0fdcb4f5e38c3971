"""Geometry of the generalised gather-convolution (segnb_conv_geom) for every convolution flavour the
reference's models use.  Pure host arithmetic, no device code.

One launch of segnb_conv_fprop computes, for (qh, qw) in [0,QH)x[0,QW):

    out[n, qh*out_step + oh0, qw*out_step + ow0, :] = sum_t in[n, qh*in_step + dh[t], qw*in_step + dw[t], :] . W[:, t, :]

Each helper returns a list of ``Launch`` (one per output parity when a stride forces it), each with
the tap table (dh, dw) and, per tap, the (kh, kw) index of the parameter tensor it reads.

  nn.Conv2d(k, stride, pad)            zf_unet.py:8, linknet.py:12,20,41,59,61, tiramisu.py:14,52,105, unet16.py:17
  its data gradient                    aten::convolution_backward (input)
  nn.ConvTranspose2d(k, stride, pad)   linknet.py:16,57, tiramisu.py:65, unet16.py:38
  its data gradient
"""
from collections import namedtuple

# taps: list of (dh, dw, kh, kw)
Launch = namedtuple('Launch', 'QH QW in_step out_step oh0 ow0 taps')


def pad8(c):
    return (int(c) + 7) // 8 * 8


def conv_out_size(size, k, stride, pad):
    return (size + 2 * pad - k) // stride + 1


def convt_out_size(size, k, stride, pad):
    return (size - 1) * stride - 2 * pad + k


def conv_fwd(Hi, Wi, kh, kw, stride, pad):
    """y[ho] = sum_k x[ho*stride - pad + k] w[k]"""
    Ho, Wo = conv_out_size(Hi, kh, stride, pad), conv_out_size(Wi, kw, stride, pad)
    taps = [(a - pad, b - pad, a, b) for a in range(kh) for b in range(kw)]
    return (Ho, Wo), [Launch(Ho, Wo, stride, 1, 0, 0, taps)]


def _scatter_phases(Hbig, Wbig, Hsmall, Wsmall, kh, kw, stride, pad):
    """big[h] = sum_{k : (h + pad - k) % stride == 0} small[(h + pad - k) / stride] w[k]
    (conv data-gradient with big = dx, small = dy; ConvTranspose forward with big = out, small = x).
    One launch per parity (ph, pw) of the big tensor; returns (launches, covers_everything)."""
    launches = []
    full = True
    for ph in range(min(stride, Hbig)):
        for pw in range(min(stride, Wbig)):
            taps = []
            for a in range(kh):
                if (ph + pad - a) % stride:
                    continue
                for b in range(kw):
                    if (pw + pad - b) % stride:
                        continue
                    taps.append(((ph + pad - a) // stride, (pw + pad - b) // stride, a, b))
            QH = (Hbig - ph + stride - 1) // stride
            QW = (Wbig - pw + stride - 1) // stride
            if not taps:
                full = False
                continue
            launches.append(Launch(QH, QW, 1, stride, ph, pw, taps))
    return launches, full


def conv_dgrad(Hi, Wi, kh, kw, stride, pad):
    """dx from dy for nn.Conv2d.  Returns (launches over the dx grid, full_coverage)."""
    Ho, Wo = conv_out_size(Hi, kh, stride, pad), conv_out_size(Wi, kw, stride, pad)
    return _scatter_phases(Hi, Wi, Ho, Wo, kh, kw, stride, pad)


def convt_fwd(Hi, Wi, kh, kw, stride, pad):
    Ho, Wo = convt_out_size(Hi, kh, stride, pad), convt_out_size(Wi, kw, stride, pad)
    launches, full = _scatter_phases(Ho, Wo, Hi, Wi, kh, kw, stride, pad)
    return (Ho, Wo), launches, full


def convt_dgrad(Hi, Wi, kh, kw, stride, pad):
    """dx[hi] = sum_k dy[hi*stride - pad + k] w[k]: a strided gather over dy, iterated over the x grid."""
    taps = [(a - pad, b - pad, a, b) for a in range(kh) for b in range(kw)]
    return [Launch(Hi, Wi, stride, 1, 0, 0, taps)]
