"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED by the reference (it has no tests);
pinned here against matplotlib-Agg renders of the reference's own draw calls (tests/golden/make_raster_golden.py).

CPU restatement (numpy float64) of what the reference's matplotlib rasterisation draws, as ordered, alpha-blended,
antialiased ORIENTED BOXES in pixel space:

  * lane way-points        ax.scatter(x, y, c=(dx, dy, 0), s=1.5, marker="D")   rasterization.py:84-91
      -> a diamond (a box turned by 45 degrees) centred on the point
  * lane segments          ax.plot(two points, linewidth=1.5, color=(dx, dy, 0)) rasterization.py:92-99
      -> a box of width 1.5 pt along the segment, lengthened by half a width at both ends (projecting caps)
  * agent boxes            plt.Rectangle(..., linewidth=1, facecolor=edgecolor=(0, 0, v), rotate_around(centre, heading))
                           visualization.py:283-289 -> the box grown by half a point on every side (its stroke)
  * canvas                 figsize = res/dpi, xlim = ylim = (-range, range), no margins: px = (x + R) / 2R * W,
                           py = (R - y) / 2R * H (row 0 is the top)                 rasterization.py:113-126
  * coverage of a pixel    product over the box's two axes of clamp(0.5 - d / w, 0, 1), d = signed distance of the
                           pixel centre to the edge pair, w = |ux| + |uy| (the width of a pixel's shadow on that axis);
                           out = out * (1 - cov) + colour * cov in draw order.
"""
import numpy as np

PT = 1.0 / 72.0  # inch


def to_pixels(xy, map_range, res):
    """World (x, y) -> pixel (px, py); rasterization.py:113-126 (equal axes, no margins, origin at the top)."""
    xy = np.asarray(xy, dtype=np.float64)
    w, h = res
    px = (xy[..., 0] + map_range) / (2.0 * map_range) * w
    py = (map_range - xy[..., 1]) / (2.0 * map_range) * h
    return np.stack([px, py], axis=-1)


def diamond_boxes(points_px, colors, s=1.5, dpi=200):
    """scatter(marker='D', s): unit square turned by 45 deg, scaled by sqrt(s) pt, stroked in the face colour.
    Half-diagonal in pixels calibrated on Agg renders at the reference's settings (s=1.5, dpi=200 -> 4.15 px: the
    2.41 px marker plus its stroke)."""
    pts = np.asarray(points_px, dtype=np.float64)
    half_diag = 4.15 * np.sqrt(s / 1.5) * dpi / 200.0
    hx = half_diag / np.sqrt(2.0)
    n = len(pts)
    c = np.sqrt(0.5)
    return np.concatenate([pts, np.full((n, 1), c), np.full((n, 1), c), np.full((n, 2), hx),
                           np.asarray(colors, dtype=np.float64).reshape(n, 3)], axis=1)


def segment_boxes(p0_px, p1_px, colors, linewidth=1.5, dpi=200):
    """plot(two points, linewidth): width linewidth pt, projecting caps (matplotlib's default solid_capstyle)."""
    p0, p1 = np.asarray(p0_px, dtype=np.float64), np.asarray(p1_px, dtype=np.float64)
    d = p1 - p0
    ln = np.linalg.norm(d, axis=1)
    u = np.where(ln[:, None] > 0, d / np.maximum(ln, 1e-30)[:, None], np.array([[1.0, 0.0]]))
    hw = 0.5 * linewidth * PT * dpi
    n = len(p0)
    return np.concatenate([(p0 + p1) / 2, u, (ln / 2 + hw)[:, None], np.full((n, 1), hw),
                           np.asarray(colors, dtype=np.float64).reshape(n, 3)], axis=1)


def agent_boxes(centres_px, lengths_px, widths_px, headings, colors, linewidth=1.0, dpi=200):
    """Rectangle(centre - size/2, length, width, linewidth=1, face = edge colour) turned by `heading` about its centre
    (counter-clockwise in world axes = clockwise on the canvas, whose y axis points down)."""
    grow = 0.5 * linewidth * PT * dpi
    h = np.asarray(headings, dtype=np.float64)
    n = len(h)
    return np.concatenate([np.asarray(centres_px, dtype=np.float64), np.cos(h)[:, None], -np.sin(h)[:, None],
                           (np.asarray(lengths_px) / 2 + grow)[:, None], (np.asarray(widths_px) / 2 + grow)[:, None],
                           np.asarray(colors, dtype=np.float64).reshape(n, 3)], axis=1)


def rasterize(boxes, res, background):
    """boxes [N][9] = (cx, cy, ux, uy, hx, hy, r, g, b) in draw order -> image [3][H][W] float64."""
    w, h = res
    out = np.empty((3, h, w), dtype=np.float64)
    out[:] = np.asarray(background, dtype=np.float64)[:, None, None]
    ys, xs = np.mgrid[0:h, 0:w]
    xs = xs + 0.5
    ys = ys + 0.5
    for cx, cy, ux, uy, hx, hy, r, g, b in np.asarray(boxes, dtype=np.float64):
        rad = np.hypot(hx, hy) + 1.5
        x0, x1 = max(int(np.floor(cx - rad)), 0), min(int(np.ceil(cx + rad)), w)
        y0, y1 = max(int(np.floor(cy - rad)), 0), min(int(np.ceil(cy + rad)), h)
        if x0 >= x1 or y0 >= y1:
            continue
        dx, dy = xs[y0:y1, x0:x1] - cx, ys[y0:y1, x0:x1] - cy
        wdt = abs(ux) + abs(uy)
        ca = np.clip(0.5 - (np.abs(dx * ux + dy * uy) - hx) / wdt, 0.0, 1.0)
        cb = np.clip(0.5 - (np.abs(-dx * uy + dy * ux) - hy) / wdt, 0.0, 1.0)
        cov = ca * cb
        for k, col in enumerate((r, g, b)):
            out[k, y0:y1, x0:x1] = out[k, y0:y1, x0:x1] * (1.0 - cov) + col * cov
    return out


def segments_hit_box(seg_p0, seg_p1, centre, length, width, heading):
    """MultiLineString(lines).intersects(rotated rectangle) (visualization.py:254-281) for one box: does any segment
    touch the closed rectangle?  Exact (slab clipping in the box frame), float64."""
    c, s = np.cos(heading), np.sin(heading)
    rot = np.array([[c, s], [-s, c]])
    a = (np.asarray(seg_p0, dtype=np.float64) - centre) @ rot.T
    b = (np.asarray(seg_p1, dtype=np.float64) - centre) @ rot.T
    d = b - a
    t0, t1 = np.zeros(len(a)), np.ones(len(a))
    ok = np.ones(len(a), dtype=bool)
    for k, half in ((0, length / 2.0), (1, width / 2.0)):
        par = d[:, k] == 0
        ok &= ~(par & (np.abs(a[:, k]) > half))
        with np.errstate(divide="ignore", invalid="ignore"):
            ta = np.where(par, -np.inf, (-half - a[:, k]) / d[:, k])
            tb = np.where(par, np.inf, (half - a[:, k]) / d[:, k])
        t0 = np.maximum(t0, np.minimum(ta, tb))
        t1 = np.minimum(t1, np.maximum(ta, tb))
    return bool(np.any(ok & (t0 <= t1)))
