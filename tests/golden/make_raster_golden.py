"""Golden rasters for row f4: a seeded synthetic scene drawn with matplotlib-Agg through the SAME draw calls the
reference makes (DriveSceneGen/utils/datasets/rasterization.py:57-126 for the lane way-points / segments and the
canvas set-up, DriveSceneGen/utils/datasets/visualization.py:283-296 for the agent rectangles).  The reference
functions themselves cannot be imported here (they need torchvision / shapely / sklearn), so the calls are restated
on plain arrays.  Run in this container:  python tests/golden/make_raster_golden.py  -> tests/golden/raster_golden.npz
"""
import io
import os

import matplotlib
matplotlib.use("Agg")
import matplotlib as mpl
import matplotlib.pyplot as plt
import numpy as np
from PIL import Image

RES, DPI, RANGE = (512, 512), 200, 100.0


def scene(seed=14555):
    """~24 lane polylines of 100 way-points (x, y, dx, dy) with dx, dy in [0, 0.99], 10 agent boxes."""
    rng = np.random.default_rng(seed)
    lanes = []
    for _ in range(24):
        p = rng.uniform(-90, 90, 2)
        th = rng.uniform(0, 2 * np.pi)
        curv = rng.uniform(-0.01, 0.01)
        pts = []
        for _ in range(100):
            pts.append(p.copy())
            p = p + 1.0 * np.array([np.cos(th), np.sin(th)])
            th += curv
        pts = np.array(pts)
        col = rng.uniform(0, 0.99, 2)
        lanes.append(np.concatenate([pts, np.tile(col, (100, 1))], axis=1))
    lanes = np.array(lanes)                                   # [24][100][4]
    agents = np.concatenate([rng.uniform(-80, 80, (10, 2)), rng.uniform(3.5, 6.0, (10, 1)), rng.uniform(1.6, 2.4, (10, 1)),
                             rng.uniform(-np.pi, np.pi, (10, 1)), rng.uniform(0, 20, (10, 1)) / 60 + 0.5], axis=1)
    return lanes, agents                                      # agents: cx, cy, length, width, heading, blue


def _finish(ax, face):
    ax.set_facecolor(np.array(face))
    ax.axis("equal")
    ax.set(xlim=(-RANGE, RANGE), ylim=(-RANGE, RANGE))
    ax.axes.get_yaxis().set_visible(False)
    ax.axes.get_xaxis().set_visible(False)
    for k in ("top", "bottom", "left", "right"):
        ax.spines[k].set_visible(False)
    plt.subplots_adjust(top=1, bottom=0, right=1, left=0, hspace=0, wspace=0)
    plt.margins(0, 0)
    buf = io.BytesIO()
    plt.savefig(buf, format="png")
    plt.close()
    buf.seek(0)
    return np.asarray(Image.open(buf).convert("RGB"))


def draw_lanes(lanes, scatter_as_line):
    fig, ax = plt.subplots(1, 1, figsize=(RES[0] / DPI, RES[1] / DPI), dpi=DPI)
    for pl in lanes:
        colors = np.concatenate([pl[:, 2:4], np.zeros((len(pl), 1))], axis=1)
        if scatter_as_line:
            ax.scatter(pl[:, 0], pl[:, 1], c=colors, s=1.5, marker="D")
        else:
            for i in range(len(pl) - 1):
                ax.plot(pl[i:i + 2, 0], pl[i:i + 2, 1], linewidth=1.5, color=colors[i])
    return _finish(ax, [0.5, 0.5, 0.5])


def draw_agents(agents):
    fig, ax = plt.subplots(1, 1, figsize=(RES[0] / DPI, RES[1] / DPI), dpi=DPI)
    for cx, cy, ln, wd, hd, blue in agents:
        col = np.array([0.0, 0.0, blue])
        ax.add_patch(plt.Rectangle((cx - ln / 2, cy - wd / 2), ln, wd, linewidth=1, facecolor=col, edgecolor=col,
                                   transform=mpl.transforms.Affine2D().rotate_around(cx, cy, hd) + ax.transData))
    ax.set_aspect("equal")
    return _finish(ax, [0.0, 0.0, 0.0])


if __name__ == "__main__":
    lanes, agents = scene()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raster_golden.npz")
    np.savez_compressed(out, lanes=lanes, agents=agents, scatter=draw_lanes(lanes, True), lines=draw_lanes(lanes, False),
                        boxes=draw_agents(agents), res=np.array(RES), dpi=np.array(DPI), map_range=np.array(RANGE))
    print("wrote", out, os.path.getsize(out), "bytes")
