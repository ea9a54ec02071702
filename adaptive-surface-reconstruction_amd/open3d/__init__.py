"""Minimal `open3d` namespace that provides only open3d.ml.torch.{ops,layers} -- the op boundary
consumed by models/v0/net_definitions_torch.py and models/common_torch.py of the reference.
GPU tensors are served by libasr_hip.so; there is no CPU implementation in this package."""
