"""Deterministic synthetic skins (the reference ships no skin PNG and there is no network).

Definition from SURVEY.md §8(d): 64x64 (S64) or 64x32 (S32) RGBA8 from the LCG
``s = s*1664525 + 1013904223 (mod 2^32)``, seed 12345, row-major; per texel r, g, b = ``s >> 24``
after one step each.  Alpha is 255 on inner-layer areas; on the outer-layer areas of the 64x64
layout (y<16 & x>=32, 32<=y<48, y>=48 & (x<16 | x>=48)) alpha = 255 if ``(s >> 28) < 6`` after a
fourth step, else 0 (~37 % opaque) -> 12 meshes.  For the legacy 64x32 layout only the head
overlay (y<16 & x>=32) is an outer area -> 7 meshes.
"""
from __future__ import annotations

import numpy as np


def synthetic_skin(kind: str = "S64", seed: int = 12345) -> np.ndarray:
    """Returns (H, 64, 4) uint8."""
    if kind not in ("S64", "S32"):
        raise ValueError("kind must be 'S64' or 'S32'")
    h = 64 if kind == "S64" else 32
    img = np.zeros((h, 64, 4), dtype=np.uint8)
    s = seed & 0xFFFFFFFF
    for y in range(h):
        for x in range(64):
            rgb = []
            for _ in range(3):
                s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
                rgb.append(s >> 24)
            if kind == "S64":
                outer = (y < 16 and x >= 32) or (32 <= y < 48) or (y >= 48 and (x < 16 or x >= 48))
            else:
                outer = y < 16 and x >= 32
            a = 255
            if outer:
                s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
                a = 255 if (s >> 28) < 6 else 0
            img[y, x] = (rgb[0], rgb[1], rgb[2], a)
    return img
