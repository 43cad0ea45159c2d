"""Numpy restatement of the first vectorisation stage of DriveSceneGen -- TEST INFRASTRUCTURE (oracle).

Follows /root/reference/DriveSceneGen/vectorization/utils/image_utils.py:6-43 (``combine_dx_dy``,
``get_gray_image``) and /root/reference/DriveSceneGen/vectorization/direct/extract_vehicles.py:136-148 (agent
channel -> uint8 -> BGR2GRAY of three equal channels -> threshold 100).  Same float operations as the
reference (float64 for the mask, float32 for the agent channel); only the per-pixel Python loop of
image_utils.py:40 is vectorised.  Parity unpinned by the reference (no tests there).
"""
import numpy as np


def get_gray_mask(img_uint8: np.ndarray) -> np.ndarray:
    """image_utils.py:13-41 -> the [H, W] uint8 mask (the reference then stacks it 3x into a PIL image)."""
    img_tensor = np.array(img_uint8, dtype=float)
    x = (img_tensor[:, :, 0] / 255.0).flatten()
    y = (img_tensor[:, :, 1] / 255.0).flatten()
    hx, bx = np.histogram(x, bins=256, range=(0, 1))
    hy, by = np.histogram(y, bins=256, range=(0, 1))
    mx, my = bx[np.argmax(hx)], by[np.argmax(hy)]
    bg = (np.fabs(x - mx) <= 0.1) & (np.fabs(y - my) <= 0.1)
    return np.where(bg, 0, 255).astype(np.uint8).reshape(img_tensor.shape[:2])


def agent_threshold(raw_img_chw_float32: np.ndarray) -> np.ndarray:
    """extract_vehicles.py:136-148 up to cv2.threshold: channel 2 -> (x*255).astype(uint8) -> gray (equal channels:
    cv2's fixed-point BGR2GRAY returns the value itself) -> > 100 ? 255 : 0."""
    ch = np.asarray(raw_img_chw_float32, dtype=np.float32)[2]
    img = (ch * 255).astype(np.uint8)
    return np.where(img > 100, 255, 0).astype(np.uint8)
