"""`inversion.resnet` of the reference (inversion/resnet.py:57): see training/face_parsing.py."""

from training.face_parsing import BasicBlock, Resnet18  # noqa: F401
