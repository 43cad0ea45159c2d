"""Direct scene rasteriser: the BEV raster the reference produces with matplotlib
(DriveSceneGen/utils/datasets/rasterization.py:15-187 `rasterize_static_map`,
DriveSceneGen/utils/datasets/visualization.py:172-330 `plot_dynamic_objects_v2`), drawn on the GPU by
`dsg_rasterize_boxes` from the same inputs: normalised, ego-aligned lane polylines `[P][100][>=5]` (x, y, _, dx, dy)
with their point masks, and agent boxes.

Every matplotlib primitive becomes an antialiased oriented box in pixel space (raster.hip); this module is the host
side: world -> pixel transform of the reference's canvas (figsize = res/dpi, limits +-map_range, no margins), the
footprint of each primitive (scatter diamond, capped 1.5-pt segment, stroked rectangle) and the reference's rule
that an agent is drawn only when its rectangle touches a lane centre line.
"""
import numpy as np
import torch

from . import _lib

_PT = 1.0 / 72.0


def _to_pixels(xy, map_range, res):
    w, h = res
    xy = np.asarray(xy, dtype=np.float64)
    return np.stack([(xy[..., 0] + map_range) / (2.0 * map_range) * w, (map_range - xy[..., 1]) / (2.0 * map_range) * h],
                    axis=-1)


def lane_boxes(polylines, masks, map_range=100.0, res=(512, 512), dpi=200, scatter_as_line=True, scatter_size=1.5,
               linewidth=1.5):
    """Boxes of the lane layer, in the reference's draw order (rasterization.py:60-99): polyline by polyline, its
    valid way-points as diamonds coloured (dx, dy, 0) -- or, with scatter_as_line=False, consecutive valid points as
    segments coloured by their first point."""
    out = []
    for pl, m in zip(polylines, masks):
        pts = np.asarray(pl, dtype=np.float64)[np.asarray(m, dtype=bool)]
        if len(pts) == 0:
            continue
        col = np.concatenate([pts[:, 3:5], np.zeros((len(pts), 1))], axis=1)
        px = _to_pixels(pts[:, :2], map_range, res)
        if scatter_as_line:
            hx = 4.15 * np.sqrt(scatter_size / 1.5) * dpi / 200.0 / np.sqrt(2.0)   # calibrated on Agg (oracle note)
            c = np.sqrt(0.5)
            out.append(np.concatenate([px, np.full((len(px), 2), c), np.full((len(px), 2), hx), col], axis=1))
        elif len(px) > 1:
            p0, p1 = px[:-1], px[1:]
            d = p1 - p0
            ln = np.linalg.norm(d, axis=1)
            u = np.where(ln[:, None] > 0, d / np.maximum(ln, 1e-30)[:, None], np.array([[1.0, 0.0]]))
            hw = 0.5 * linewidth * _PT * dpi
            out.append(np.concatenate([(p0 + p1) / 2, u, (ln / 2 + hw)[:, None], np.full((len(p0), 1), hw), col[:-1]],
                                      axis=1))
    return np.concatenate(out, axis=0) if out else np.zeros((0, 9))


def agents_on_lanes(agents, polylines, masks):
    """visualization.py:254-281: an agent rectangle is drawn only if it touches one of the direction lines (the lane
    polylines whose 100 points are all valid, rasterization.py:101-110).  agents [A][6] = (x, y, length, width,
    heading, blue).  Exact segment / rectangle test in the rectangle's frame."""
    segs0, segs1 = [], []
    for pl, m in zip(polylines, masks):
        if np.all(m):
            p = np.asarray(pl, dtype=np.float64)[:, :2]
            segs0.append(p[:-1])
            segs1.append(p[1:])
    if not segs0:
        return np.zeros(len(agents), dtype=bool)
    a0, a1 = np.concatenate(segs0), np.concatenate(segs1)
    keep = np.zeros(len(agents), dtype=bool)
    for k, (cx, cy, ln, wd, hd, _) in enumerate(np.asarray(agents, dtype=np.float64)):
        c, s = np.cos(hd), np.sin(hd)
        rot = np.array([[c, s], [-s, c]])
        a = (a0 - (cx, cy)) @ rot.T
        d = (a1 - (cx, cy)) @ rot.T - a
        t0, t1 = np.zeros(len(a)), np.ones(len(a))
        ok = np.ones(len(a), dtype=bool)
        for ax, half in ((0, ln / 2.0), (1, wd / 2.0)):
            par = d[:, ax] == 0
            ok &= ~(par & (np.abs(a[:, ax]) > half))
            with np.errstate(divide="ignore", invalid="ignore"):
                ta = np.where(par, -np.inf, (-half - a[:, ax]) / d[:, ax])
                tb = np.where(par, np.inf, (half - a[:, ax]) / d[:, ax])
            t0 = np.maximum(t0, np.minimum(ta, tb))
            t1 = np.minimum(t1, np.maximum(ta, tb))
        keep[k] = bool(np.any(ok & (t0 <= t1)))
    return keep


def agent_boxes(agents, map_range=100.0, res=(512, 512), dpi=200, linewidth=1.0):
    """Rectangle((x - l/2, y - w/2), l, w, linewidth=1, face = edge = (0, 0, blue)) turned by the heading about its
    centre (visualization.py:283-289): the box grown by half the stroke on every side."""
    a = np.asarray(agents, dtype=np.float64).reshape(-1, 6)
    sx = res[0] / (2.0 * map_range)
    grow = 0.5 * linewidth * _PT * dpi
    z = np.zeros(len(a))
    return np.stack([*_to_pixels(a[:, :2], map_range, res).T, np.cos(a[:, 4]), -np.sin(a[:, 4]),
                     a[:, 2] * sx / 2 + grow, a[:, 3] * sx / 2 + grow, z, z, a[:, 5]], axis=1)


def rasterize_boxes(boxes, res, background, device="cuda"):
    """dsg_rasterize_boxes: ordered box list [N][9] -> [3][H][W] fp32 on the GPU."""
    b = torch.as_tensor(np.ascontiguousarray(boxes, dtype=np.float32)).to(device)
    w, h = res
    out = torch.empty((3, h, w), dtype=torch.float32, device=b.device)
    with torch.cuda.device(b.device):
        _lib.check(_lib.load().dsg_rasterize_boxes(_lib.ptr(b) if len(b) else None, len(b), _lib.ptr(out), h, w,
                                                   float(background[0]), float(background[1]), float(background[2]),
                                                   torch.cuda.current_stream().cuda_stream))
    return out


def rasterize_static_map(polylines, masks, agents=None, img_res=(512, 512), dpi=200, map_range=100.0, with_agent=False,
                         scatter_as_line=True, scatter_size=1.5, device="cuda"):
    """The reference's `rasterize_static_map(...)` tensor branch (rasterization.py:150-187): `[H][W][3]` fp32 in [0, 1]
    -- lane layer on 0.5 grey; with_agent replaces the blue channel by the agent layer's (black canvas, rectangles
    coloured by speed, only agents standing on a lane centre line)."""
    lanes = rasterize_boxes(lane_boxes(polylines, masks, map_range, img_res, dpi, scatter_as_line, scatter_size),
                            img_res, (0.5, 0.5, 0.5), device)
    if with_agent:
        ag = np.asarray(agents, dtype=np.float64).reshape(-1, 6)
        ag = ag[agents_on_lanes(ag, polylines, masks)]
        traj = rasterize_boxes(agent_boxes(ag, map_range, img_res, dpi), img_res, (0.0, 0.0, 0.0), device)
        lanes = torch.cat([lanes[:2], traj[2:3]], dim=0)
    return lanes.permute(1, 2, 0)
