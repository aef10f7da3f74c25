"""`inversion.BiSeNet` of the reference (inversion/BiSeNet.py:229): the module tree lives in training/face_parsing.py (same parameter names:
`segNet-20Class.pth` loads unchanged, dnnlib/seg_tools.py:128)."""

from training.face_parsing import (AttentionRefinementModule, BiSeNet, BiSeNetOutput, ContextPath, ConvBNReLU,  # noqa: F401
                                   FeatureFusionModule)
