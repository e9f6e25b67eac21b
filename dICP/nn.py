from dicp_amd.nn import nn  # noqa: F401
