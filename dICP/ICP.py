from dicp_amd.ICP import ICP  # noqa: F401
