"""segnb -- host side of the MI355X-native segmentation hot path.

    _native   ctypes binding of libsegnb_hip.so (include/segnb_hip.h); no fallback
    convplan  tap tables of the generalised gather-convolution
    engine    NHWC views, packed-weight conv ops, fused conv->BN->act stages, flat params
    seglosses loss / metric autograd functions over the fused loss kernels
    dist      data-parallel gradient all-reduce over RCCL (torch.distributed backend "nccl")
"""
__version__ = '0.1.0'
