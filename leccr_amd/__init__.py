"""leccr_amd - MI355X-native (gfx950) implementation of the LECCR retrieval hot path.

Python host code mirroring the reference's models/ API over a C-ABI HIP library (libleccr_hip.so).
There is no CPU or PyTorch-eager fallback: importing leccr_amd.ops without the built library raises.
"""
__version__ = '0.1.0'
