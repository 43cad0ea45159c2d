"""Row f4 (direct rasteriser): GPU kernel vs the numpy oracle, and both vs matplotlib-Agg renders of the reference's
draw calls (tests/golden/raster_golden.npz, made by tests/golden/make_raster_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "raster_golden.npz"))
RES, DPI, RANGE = tuple(int(v) for v in GOLD["res"]), int(GOLD["dpi"]), float(GOLD["map_range"])


def _oracle_layers():
    lanes, agents = GOLD["lanes"], GOLD["agents"]
    pts = ro.to_pixels(lanes[:, :, :2].reshape(-1, 2), RANGE, RES)
    cols = np.concatenate([lanes[:, :, 2:4].reshape(-1, 2), np.zeros((len(pts), 1))], axis=1)
    p0 = ro.to_pixels(lanes[:, :-1, :2].reshape(-1, 2), RANGE, RES)
    p1 = ro.to_pixels(lanes[:, 1:, :2].reshape(-1, 2), RANGE, RES)
    c2 = np.concatenate([lanes[:, :-1, 2:4].reshape(-1, 2), np.zeros((len(p0), 1))], axis=1)
    sc = RES[0] / (2 * RANGE)
    blue = np.stack([0 * agents[:, 5], 0 * agents[:, 5], agents[:, 5]], axis=1)
    return {"scatter": (ro.diamond_boxes(pts, cols, 1.5, DPI), [0.5] * 3),
            "lines": (ro.segment_boxes(p0, p1, c2, 1.5, DPI), [0.5] * 3),
            "boxes": (ro.agent_boxes(ro.to_pixels(agents[:, :2], RANGE, RES), agents[:, 2] * sc, agents[:, 3] * sc,
                                     agents[:, 4], blue, 1.0, DPI), [0.0] * 3)}


@pytest.mark.parametrize("layer", ["scatter", "lines", "boxes"])
def test_oracle_matches_matplotlib_agg(layer):
    """The box model against Agg's exact-area rasteriser: stated tolerance -- mean |err| <= 0.005 of full scale, at most
    0.5 % of the pixels off by more than 0.1, none by more than 0.3 (edge pixels of rotated shapes)."""
    boxes, bg = _oracle_layers()[layer]
    got = ro.rasterize(boxes, RES, bg)
    want = GOLD[layer].astype(np.float64).transpose(2, 0, 1) / 255.0
    d = np.abs(got - want)
    assert d.mean() <= 0.005, d.mean()
    assert (d > 0.1).mean() <= 0.005, (d > 0.1).mean()
    assert d.max() <= 0.3, d.max()


def test_segment_rectangle_test_known_answers():
    p0, p1 = np.array([[-10.0, 0.0], [0.0, 5.0]]), np.array([[10.0, 0.0], [3.0, 9.0]])
    assert ro.segments_hit_box(p0, p1, np.array([0.0, 0.0]), 4.0, 2.0, 0.3)            # crosses the centre
    assert not ro.segments_hit_box(p0[1:], p1[1:], np.array([0.0, 0.0]), 4.0, 2.0, 0.3)  # passes above
    assert ro.segments_hit_box(p0[1:], p1[1:], np.array([1.0, 6.0]), 4.0, 2.0, 0.0)    # end point inside
    assert not ro.segments_hit_box(p0[:1], p1[:1], np.array([0.0, 3.0]), 4.0, 2.0, 0.0)  # parallel, outside the slab


@pytest.mark.gpu
@pytest.mark.parametrize("layer", ["scatter", "lines", "boxes"])
def test_gpu_rasteriser_matches_oracle_and_agg(layer):
    from drivescenegen_amd import rasterization as rz
    boxes, bg = _oracle_layers()[layer]
    got = rz.rasterize_boxes(boxes, RES, bg).cpu().numpy().astype(np.float64)
    want = ro.rasterize(boxes, RES, bg)
    assert np.abs(got - want).max() <= 2e-5, np.abs(got - want).max()   # fp32 kernel vs float64 oracle
    agg = GOLD[layer].astype(np.float64).transpose(2, 0, 1) / 255.0
    assert np.abs(got - agg).mean() <= 0.005


@pytest.mark.gpu
def test_gpu_rasterize_static_map_layers_and_agent_rule():
    """Host side: draw order / masks / the lanes-only-under-agents rule, against the oracle's builders."""
    from drivescenegen_amd import rasterization as rz
    lanes, agents = GOLD["lanes"], GOLD["agents"].copy()
    agents[:3, :2] = lanes[[0, 5, 12], 50, :2]   # three agents standing on a fully valid lane,
    agents[3, :2] = lanes[3, 80, :2]              # one on the masked-out tail of a partly valid one
    poly = np.concatenate([lanes[:, :, :2], np.zeros((len(lanes), 100, 1)), lanes[:, :, 2:4]], axis=2)  # x y z dx dy
    masks = np.ones(poly.shape[:2], dtype=bool)
    masks[3, 60:] = False          # a partly valid polyline: drawn, but not a direction line
    masks[7, :] = False            # an empty one
    img = rz.rasterize_static_map(poly, masks, agents, RES, DPI, RANGE, with_agent=True).cpu().numpy()
    assert img.shape == (RES[1], RES[0], 3)
    # lanes layer (R, G) == oracle on the valid points, in polyline order
    pts, cols = [], []
    for pl, m in zip(poly, masks):
        v = pl[m]
        pts.append(ro.to_pixels(v[:, :2], RANGE, RES))
        cols.append(np.concatenate([v[:, 3:5], np.zeros((len(v), 1))], axis=1))
    want = ro.rasterize(ro.diamond_boxes(np.concatenate(pts), np.concatenate(cols), 1.5, DPI), RES, [0.5] * 3)
    assert np.abs(img[:, :, :2].transpose(2, 0, 1) - want[:2]).max() <= 2e-5
    # agent layer (B): only agents whose rectangle touches a fully valid polyline
    segs0 = np.concatenate([pl[:-1, :2] for pl, m in zip(poly, masks) if m.all()])
    segs1 = np.concatenate([pl[1:, :2] for pl, m in zip(poly, masks) if m.all()])
    keep = np.array([ro.segments_hit_box(segs0, segs1, a[:2], a[2], a[3], a[4]) for a in agents])
    assert np.array_equal(keep, rz.agents_on_lanes(agents, poly, masks)) and keep[:3].all() and not keep.all()
    ag = agents[keep]
    sc = RES[0] / (2 * RANGE)
    blue = np.stack([0 * ag[:, 5], 0 * ag[:, 5], ag[:, 5]], axis=1)
    wantb = ro.rasterize(ro.agent_boxes(ro.to_pixels(ag[:, :2], RANGE, RES), ag[:, 2] * sc, ag[:, 3] * sc, ag[:, 4], blue,
                                        1.0, DPI), RES, [0.0] * 3)
    assert np.abs(img[:, :, 2] - wantb[2]).max() <= 2e-5
    # no agents / no boxes at all: plain canvases
    empty = rz.rasterize_boxes(np.zeros((0, 9)), (64, 48), (0.25, 0.5, 0.75)).cpu()
    assert empty.shape == (3, 48, 64) and torch.all(empty[1] == 0.5)
