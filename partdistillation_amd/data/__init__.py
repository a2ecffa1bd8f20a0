from .device_mapper import DeviceProposalMapper  # noqa: F401
