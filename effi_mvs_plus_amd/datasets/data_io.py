"""PFM files and line reading (reference: datasets/data_io.py:18-126).  Same names, arguments, return values and exceptions."""
import re
import sys

import numpy as np


def read_all_lines(filename):
    """reference: datasets/data_io.py:18-21"""
    with open(filename) as f:
        return [line.rstrip() for line in f.readlines()]


def read_pfm(filename):
    """-> (data [h,w] or [h,w,3] float32, rows flipped to top-down; scale > 0).  reference: datasets/data_io.py:61-95"""
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header == "PF":
            color = True
        elif header == "Pf":
            color = False
        else:
            raise Exception("Not a PFM file.")
        dim_match = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not dim_match:
            raise Exception("Malformed PFM header.")
        width, height = map(int, dim_match.groups())
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"          # negative scale marks little-endian samples
        scale = abs(scale)
        data = np.fromfile(f, endian + "f")
    shape = (height, width, 3) if color else (height, width)
    return np.flipud(np.reshape(data, shape)), scale


pfm_imread = read_pfm                                # reference: datasets/data_io.py:25-59 (same function, file left open)


def save_pfm(filename, image, scale=1):
    """image: float32 [h,w], [h,w,1] or [h,w,3]; rows are written bottom-up.  reference: datasets/data_io.py:98-126"""
    image = np.flipud(image)
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if len(image.shape) == 3 and image.shape[2] == 3:
        color = True
    elif len(image.shape) == 2 or len(image.shape) == 3 and image.shape[2] == 1:
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    endian = image.dtype.byteorder
    if endian == "<" or endian == "=" and sys.byteorder == "little":
        scale = -scale
    with open(filename, "wb") as f:
        f.write(("PF\n" if color else "Pf\n").encode("utf-8"))
        f.write("{} {}\n".format(image.shape[1], image.shape[0]).encode("utf-8"))
        f.write(("%f\n" % scale).encode("utf-8"))
        image.tofile(f)
