#!/usr/bin/env python3
"""Writes assets/templates/*.png and tests/golden/templates.json.

The pixel grids are the decoded content of the reference's template/{2x2,3x3,4x4}-01.png
(SURVEY.md Appendix C).  When /root/reference is present the grids are re-checked against a PIL
decode of those files; the PNGs written here are produced by PIL from the grids (data, not a copy
of the reference's files)."""
import json, os, sys
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRIDS = {
    "2x2-01": [[0,0,0,0],[0,1,0,0],[0,0,1,0],[0,0,0,0]],
    "3x3-01": [[0,0,0,0,0],[0,1,1,1,0],[0,1,1,0,0],[0,1,0,1,0],[0,0,0,0,0]],
    "4x4-01": [[0,0,0,0,0,0],[0,1,0,1,1,0],[0,0,1,1,1,0],[0,0,1,1,1,0],[0,1,0,1,1,0],[0,0,0,0,0,0]],
}
# SURVEY.md Appendix C, column (A): codes under the stride quirk with zero padding
CODES = {"2x2-01": [0x8,0x2,0x1,0x4], "3x3-01": [0x174,0x117,0x5d,0x1d1], "4x4-01": [0xdeed,0x96ff,0xb77b,0xff69]}

def main():
    os.makedirs(os.path.join(ROOT, "assets/templates"), exist_ok=True)
    out = {}
    for name, g in GRIDS.items():
        a = (np.array(g, dtype=np.uint8) * 255)
        ref = f"/root/reference/template/{name}.png"
        if os.path.exists(ref):
            r = np.array(Image.open(ref).convert("L"))
            assert (r == a).all(), name
        Image.fromarray(a, "L").save(os.path.join(ROOT, f"assets/templates/{name}.png"))
        out[name] = {"pixels": a.tolist(), "codes": CODES[name]}
    with open(os.path.join(ROOT, "tests/golden/templates.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("ok")
if __name__ == "__main__":
    main()
