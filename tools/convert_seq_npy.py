#!/usr/bin/env python3
"""One-off conversion of the reference's ImageNet-VID sequence lists (object-array .npy, yolox/data/datasets/vid.py:48-75 of the
reference) to the .json list VIDDataset reads by default.  Reading such a .npy unpickles it, i.e. runs what the file says -- do this
once, by hand, on a file you made yourself with the reference's tooling.

    python tools/convert_seq_npy.py yolox/data/datasets/train_seq.npy [out.json]
"""
import json
import os
import sys

import numpy as np


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.splitext(src)[0] + ".json"
    videos = [[str(f) for f in v] for v in np.load(src, allow_pickle=True).tolist()]
    with open(dst, "w") as fh:
        json.dump(videos, fh)
    print(f"{src}: {len(videos)} videos, {sum(len(v) for v in videos)} frames -> {dst}")


if __name__ == "__main__":
    main()
