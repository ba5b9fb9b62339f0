"""A SECOND, independently structured restatement of COCO's run-length mask format.  TEST INFRASTRUCTURE ONLY.

Why: pycocotools (cocoapi `common/maskApi.c`: rleEncode / rleToString / rleFrString / rleDecode) is absent from this image
and from /root/reference, so `oracle.ops.rle_*` -- the restatement the HIP kernels are held to -- cannot be pinned against
the library itself (PARITY UNPINNED).  What CAN be excluded is a slip of the restatement: the functions
here state the same published format in a different shape -- a pixel-by-pixel column scan instead of numpy edge
differences, a CLOSED FORM for the string digits (signed base-32, minimal length) instead of maskApi.c's shift loop, a
digit-sum parser instead of the bit-or loop -- and tests/test_oracle_ops.py holds the two to each other on random and
adversarial masks plus a few answers worked by hand from the format's definition.

The format (cocoapi, maskApi.h): a binary mask [H,W] is scanned in COLUMN-major order; `counts` are the lengths of the
alternating runs 0,1,0,1,... starting with zeros (a mask whose first pixel is 1 starts with a zero-length run).  The
compressed string stores counts[i] for i <= 2 and counts[i] - counts[i-2] for i > 2 (runs of the same colour tend to
repeat) as signed integers in 5-bit groups, least significant group first, each group one character chr(48 + group +
32 * more_groups_follow); the sign is the top bit (value 16) of the last group, and the number of groups is minimal."""
import numpy as np


def counts_by_scan(mask):
    m = np.asarray(mask)
    h, w = m.shape
    runs, colour, length = [], 0, 0
    for x in range(w):                    # column-major: all rows of column 0, then column 1, ...
        col = m[:, x].tolist()
        for v in col:
            v = 1 if v else 0
            if v == colour:
                length += 1
            else:
                runs.append(length)
                colour, length = v, 1
    runs.append(length)
    return runs


def _digits(x):
    """minimal signed base-32 representation of x, least significant first: n groups with -2^(5n-1) <= x < 2^(5n-1)"""
    mag = x if x >= 0 else -x - 1                    # (~x for negatives: both need bit_length(mag) <= 5n - 1)
    n = max(1, -(-(mag.bit_length() + 1) // 5))
    u = x % (1 << (5 * n))                           # two's complement in 5n bits
    return [(u >> (5 * i)) & 31 for i in range(n)]


def string_by_signed_groups(cnts):
    out = bytearray()
    for i, c in enumerate(cnts):
        x = int(c) - (int(cnts[i - 2]) if i >= 3 else 0)
        d = _digits(x)
        for k, g in enumerate(d):
            out.append(48 + g + (32 if k + 1 < len(d) else 0))
    return bytes(out)


def parse_string(s):
    vals, group = [], []
    for ch in bytes(s):
        c = ch - 48
        group.append(c & 31)
        if not (c & 32):                             # last group of this value
            x = sum(g << (5 * k) for k, g in enumerate(group))
            if group[-1] & 16:
                x -= 1 << (5 * len(group))
            group = []
            if len(vals) >= 3:
                x += vals[-2]
            vals.append(x)
    assert not group, "string ends inside a value"
    return vals


def decode_by_columns(cnts, h, w):
    flat, colour = [], 0
    for c in cnts:
        flat.extend([colour] * int(c))
        colour ^= 1
    assert len(flat) == h * w, (len(flat), h, w)
    return np.array(flat, np.uint8).reshape((h, w), order="F")
