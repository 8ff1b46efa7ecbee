"""Drop-in for multiframe/data/optical_flow/model/correlation_package/correlation.py (the one
native extension of the reference tree): `Correlation(pad_size, kernel_size, max_displacement,
stride1, stride2, corr_multiply)` with the configuration MaskFlownet uses (MaskFlownet.py:116, 416:
pad_size = max_displacement, kernel_size = 1, strides 1).  Forward only: ACFM never trains the flow
network, the reference's backward kernels are never executed."""
from torch import nn

from . import ops


class Correlation(nn.Module):
    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        if kernel_size != 1 or stride1 != 1 or stride2 != 1 or pad_size != max_displacement or corr_multiply != 1 \
                or not 1 <= max_displacement <= 4:
            raise NotImplementedError(
                "Correlation: only MaskFlownet's configuration is built (kernel_size=1, stride1=stride2=1, "
                "pad_size=max_displacement in 1..4, corr_multiply=1)")
        self.pad_size, self.kernel_size, self.max_displacement = pad_size, kernel_size, max_displacement
        self.stride1, self.stride2, self.corr_multiply = stride1, stride2, corr_multiply

    def forward(self, input1, input2):
        return ops.correlation(input1, input2, self.max_displacement)
