from .ms_deform_attn import MSDeformAttn  # noqa: F401
