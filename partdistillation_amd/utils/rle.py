"""COCO run-length encoding of binary masks (the on-disk pseudo-label format of the reference:
utils/utils.py:15-32 proposals_to_coco_json -> pycocotools mask_util.encode, consumed by the dataset mappers).

pycocotools (2.0.x, third party, absent here) stores a mask as {"size": [h, w], "counts": <string>}: run lengths of the
COLUMN-major flattened mask, starting with a run of zeros, each count c_i (c_i - c_{i-2} for i > 2) written in 5-bit
groups, low group first, bit 0x20 = "more groups follow", bit 0x10 of the last group = sign, + 48 to land in ASCII
(maskApi.c rleToString / rleFrString).  Restated here from that published format; `decode` is its inverse."""
import numpy as np


def mask_to_counts(mask: np.ndarray) -> np.ndarray:
    """mask [h, w] (bool / 0-1) -> run lengths of the column-major flattening, first run = zeros (possibly empty)."""
    m = np.asarray(mask).astype(bool).T.reshape(-1)                          # column-major order
    if m.size == 0:
        return np.zeros((0,), dtype=np.int64)
    change = np.flatnonzero(m[1:] != m[:-1]) + 1
    bounds = np.concatenate(([0], change, [m.size]))
    counts = np.diff(bounds)
    if m[0]:
        counts = np.concatenate(([0], counts))
    return counts.astype(np.int64)


def counts_to_string(counts) -> bytes:
    """vectorised rleToString: all counts advance one 5-bit group per pass (at most 13 passes for int64)"""
    c = np.asarray(counts, dtype=np.int64).reshape(-1)
    n = c.shape[0]
    if n == 0:
        return b""
    x = c.copy()
    if n > 3:
        x[3:] -= c[1:-2]
    active = np.ones(n, dtype=bool)
    cols, valids = [], []
    while active.any():
        ch = x & 0x1F
        x = x >> 5                                                     # arithmetic shift
        more = np.where((ch & 0x10) != 0, x != -1, x != 0)
        cols.append(((ch | np.where(more, 0x20, 0)) + 48).astype(np.uint8))
        valids.append(active.copy())
        active &= more
    chars = np.stack(cols, axis=1)                                     # [n, groups], row-major = emission order
    return chars[np.stack(valids, axis=1)].tobytes()


def string_to_counts(s) -> np.ndarray:
    if isinstance(s, str):
        s = s.encode("ascii")
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return np.asarray(cnts, dtype=np.int64)


def encode(mask: np.ndarray) -> dict:
    h, w = mask.shape
    return {"size": [int(h), int(w)], "counts": counts_to_string(mask_to_counts(mask))}


def decode(rle: dict) -> np.ndarray:
    h, w = rle["size"]
    counts = string_to_counts(rle["counts"])
    flat = np.zeros(h * w, dtype=bool)
    pos, val = 0, False
    for c in counts:
        if val:
            flat[pos:pos + c] = True
        pos += int(c)
        val = not val
    return flat.reshape(w, h).T


def labels_to_coco_json(labels: np.ndarray, present) -> list:
    """label map [h, w] (0 = background) -> the reference's list[{"segmentation": rle}] (counts as utf-8 str), one entry
    per label in `present` (ascending), like proposals_to_coco_json(binary_mask) on the stacked per-label masks."""
    out = []
    for l in present:
        rle = encode(labels == l)
        rle["counts"] = rle["counts"].decode("utf-8")
        out.append({"segmentation": rle})
    return out


def runs_to_coco_json(values: np.ndarray, lengths: np.ndarray, size, present) -> list:
    """same as labels_to_coco_json, from the run-length form of the COLUMN-major flattened label map (values[i] repeated
    lengths[i] times): the per-label masks are merges of those runs, no per-pixel work."""
    out = []
    h, w = int(size[0]), int(size[1])
    values, lengths = np.asarray(values), np.asarray(lengths, dtype=np.int64)
    ends = np.cumsum(lengths)
    for l in present:
        on = values == l
        if on.size == 0:
            counts = np.zeros((0,), dtype=np.int64)
        else:
            change = np.flatnonzero(on[1:] != on[:-1]) + 1             # first run of each merged group
            bounds = np.concatenate(([0], ends[change - 1], [ends[-1]]))
            counts = np.diff(bounds)
            if on[0]:
                counts = np.concatenate(([0], counts))
        out.append({"segmentation": {"size": [h, w], "counts": counts_to_string(counts).decode("utf-8")}})
    return out


def masks_to_coco_json(masks) -> list:
    """bool [n, h, w] (numpy or CPU tensor) -> list[{"segmentation": rle}] with utf-8 `counts`: the reference's
    utils/utils.py:15-32 proposals_to_coco_json (pycocotools encode of every mask)."""
    out = []
    for m in np.asarray(masks).astype(bool):
        r = encode(m)
        r["counts"] = r["counts"].decode("utf-8")
        out.append({"segmentation": r})
    return out

