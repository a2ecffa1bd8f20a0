"""partdistillation_amd — MI355X-native Mask2Former part-proposal /
part-distillation training step (hot path of facebookresearch/PartDistillation).

Hand-written HIP kernels for gfx950 live in ``csrc/`` behind the C-ABI of
``include/*.h`` (``libpd_hip.so``); the Python here mirrors the reference's
operator / registry interface for that path.  See DESIGN.md.
"""
__version__ = "0.1.0"
