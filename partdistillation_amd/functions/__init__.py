"""Python faces of the HIP operators of libpd_hip.so (no fallbacks)."""
