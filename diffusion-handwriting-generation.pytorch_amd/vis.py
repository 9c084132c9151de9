"""Output side of the sampler (reference utils/vis.py:5-36): offsets -> pen positions -> polylines."""
from __future__ import annotations

import numpy as np


def strokes_to_polylines(strokes: np.ndarray) -> list[np.ndarray]:
    """[L,3] = (dx, dy, pen) -> list of [n,2] position arrays, one per pen-down stretch.  Positions are the running sum of
    the offsets; a rounded pen value of 1 at row i ends the stretch BEFORE row i (the move to i is a jump), vis.py:13-31."""
    strokes = np.asarray(strokes)
    pos = np.cumsum(strokes[:, :2], axis=0)
    ends = np.flatnonzero(np.round(strokes[:, 2]) != 0)
    lines, prev = [], 0
    for ind in ends:
        lines.append(pos[prev:ind])
        prev = int(ind)
    return lines


def show_strokes(strokes: np.ndarray, name: str = "", show_output: bool = True, scale: int = 1) -> None:
    """Plot the strokes (and save ./<name>.png); needs matplotlib."""
    import matplotlib

    if not show_output:
        matplotlib.use("Agg")
    from matplotlib import pyplot as plt

    pos = np.cumsum(np.asarray(strokes)[:, :2], axis=0).T
    w, h = np.max(pos, axis=-1) - np.min(pos, axis=-1)
    plt.figure(figsize=(scale * w / h, scale))
    plt.axis("off")
    for line in strokes_to_polylines(strokes):
        plt.plot(line[:, 0], line[:, 1], color="black")
    if name:
        plt.savefig(f"./{name}.png", bbox_inches="tight")
    if show_output:
        plt.show()
    else:
        plt.close()
