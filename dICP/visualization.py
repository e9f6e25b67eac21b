from dicp_amd.visualization import plot_overlay, plot_map  # noqa: F401
