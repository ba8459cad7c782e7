"""Side-by-side scatter of a generated and a ground-truth cloud, as a uint8 image.

Mirrors reference ``src/models/visualization.py:9-28`` (``visualize_point_clouds``): same
figure (6x3 in, two 3-D axes, titles ``Sample: i`` / ``Ground Truth: i``, marker size 5),
returned as ``[4, H, W]`` RGBA like the reference's renderer buffer.  PNG writing uses PIL
(``imageio``, which the reference imports, is not needed).  Host-side only.
"""
from __future__ import annotations

import numpy as np


def visualize_point_clouds(pts, gtr, idx, pert_order=(0, 1, 2)):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    pts = pts.detach().cpu().numpy()[:, list(pert_order)]
    gtr = gtr.detach().cpu().numpy()[:, list(pert_order)]
    fig = plt.figure(figsize=(6, 3))
    for pos, cloud, title in ((121, pts, f"Sample: {idx}"), (122, gtr, f"Ground Truth: {idx}")):
        ax = fig.add_subplot(pos, projection="3d")
        ax.set_title(title)
        ax.scatter(cloud[:, 0], cloud[:, 1], cloud[:, 2], s=5)
    fig.canvas.draw()
    img = np.asarray(fig.canvas.buffer_rgba()).copy()
    plt.close(fig)
    return np.transpose(img, (2, 0, 1))


def write_png(path: str, hwc: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(hwc).astype(np.uint8)).save(path)
