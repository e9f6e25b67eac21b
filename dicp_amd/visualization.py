"""Debug scatter plots (the reference's dICP/visualization.py:5-44 interface).  Not on the
hot path; matplotlib is imported lazily so that importing this module never needs it."""
import torch


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a


def plot_overlay(pc1, pc2, c1='b', c2='r', file_name="overlay.png"):
    import matplotlib.pyplot as plt
    pc1, pc2 = _np(pc1), _np(pc2)
    plt.figure()
    plt.scatter(pc1[:, 0], pc1[:, 1], s=0.5, c=c1)
    plt.scatter(pc2[:, 0], pc2[:, 1], s=0.5, c=c2)
    plt.savefig(file_name)
    plt.close()


def plot_map(points, color='b', map=None):
    import matplotlib.pyplot as plt
    points = _np(points)
    plt.figure()
    if map is not None:
        mp = _np(map)
        plt.scatter(mp[:, 0], mp[:, 1], s=0.5, c='k')
    plt.scatter(points[:, 0], points[:, 1], s=0.5, c=color)
    plt.axis('equal')
    plt.savefig("map.png")
    plt.close()
