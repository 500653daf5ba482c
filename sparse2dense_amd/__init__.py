"""sparse2dense_amd — MI355X-native voxel-backbone hot path of Sparse2Dense (CenterPoint + S2D).

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); the compute is
hand-written HIP for gfx950 behind the C-ABI declared in include/s2d.h
(sparse2dense_amd/csrc -> sparse2dense_amd/lib/libs2d_hip.so).
"""
__version__ = "0.1.0"
