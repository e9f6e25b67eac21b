from dicp_amd.loss import loss  # noqa: F401
