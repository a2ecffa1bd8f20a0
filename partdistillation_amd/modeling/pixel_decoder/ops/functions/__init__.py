from .ms_deform_attn_func import MSDeformAttnFunction, ms_deform_attn  # noqa: F401
