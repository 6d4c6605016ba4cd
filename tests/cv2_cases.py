"""Hand-derived vectors for the contour / bounding-box step (reference clip/utils.py:115-142 and its
caller clip/clip_tool.py:179-183).  Real OpenCV is absent from the image, so the oracle and the fixtures use a stand-in
(oracle/refharness.py:31-71); these cases pin the DOCUMENTED OpenCV semantics the stand-in claims, each expected mask worked
out by hand from the reference's own lines:

  u8  = (cam * 255).astype(uint8)                                   utils.py:117  (truncation)
  thr = int(threshold * max(u8));  on = u8 > thr                     utils.py:118-122; OpenCV docs, "threshold", THRESH_BINARY:
                                                                     dst = maxval if src > thresh else 0 (strictly greater)
  contours = findContours(on, RETR_TREE, CHAIN_APPROX_SIMPLE)        utils.py:123-126; OpenCV docs, "findContours": borders of
      the non-zero regions by Suzuki-Abe border following -- 1-pixels are 8-CONNECTED (two pixels touching at a corner belong
      to one contour); RETR_TREE also returns hole borders, which run along 1-pixels of the surrounding component, i.e. inside
      that component's bounding box; no non-zero pixel -> no contour
  no contour -> [[0, 0, 0, 0]], count 1                              utils.py:128-129
  x, y, w, h = boundingRect(contour)                                 utils.py:136; OpenCV docs, "boundingRect": up-right
      rectangle of the point set, w = xmax - xmin + 1, h = ymax - ymin + 1
  box = [x, y, min(x + w, W - 1), min(y + h, H - 1)]                 utils.py:137-140
  mask[y0:y1, x0:x1] = 1   (exclusive ends)                          clip_tool.py:181-183

Consequence worth a case of its own: a box that reaches the last column / row loses it (x + w = W is clamped to W - 1 and then
used as an EXCLUSIVE end), so a component lying only in the last column leaves no mask at all.

Levels are written as integers L and turned into cam = (L + 0.5) / 255, so that the reference's truncating uint8 conversion
gives exactly L whatever the float rounding."""
import numpy as np

H, W = 8, 10


def _cam(levels):
    return ((np.asarray(levels, dtype=np.float64) + 0.5) / 255.0).astype(np.float32)


def _blank(level=0):
    return np.full((H, W), level, dtype=np.int64)


def cases():
    """-> list of (name, cam float32 (H, W), threshold, expected mask float32 (H, W))."""
    out = []

    def add(name, levels, thr, boxes):
        m = np.zeros((H, W), np.float32)
        for (y0, y1, x0, x1) in boxes:          # python slices, ends exclusive, already clamped by hand
            m[y0:y1, x0:x1] = 1
        out.append((name, _cam(levels), thr, m))

    # 1. one interior blob, rows 2..4, columns 3..6: boundingRect = (3, 2, 4, 3) -> box [3, 2, 7, 5]: exactly the blob
    L = _blank(10); L[2:5, 3:7] = 200
    add("interior blob", L, 0.4, [(2, 5, 3, 7)])
    # 2. blob reaching the last row and the last column (rows 5..7, columns 7..9): rect (7, 5, 3, 3) -> x1 = min(10, 9) = 9,
    #    y1 = min(8, 7) = 7 -> mask rows 5..6, columns 7..8: the border row / column is dropped by the exclusive end
    L = _blank(10); L[5:8, 7:10] = 200
    add("blob on the bottom-right border", L, 0.4, [(5, 7, 7, 9)])
    # 3. a ring (rows 1..5, columns 1..5 with the centre pixel off): the hole's border runs on ring pixels, so its rectangle
    #    lies inside the outer one: union = the whole 5 x 5 square including the hole pixel
    L = _blank(0); L[1:6, 1:6] = 250; L[3, 3] = 0
    add("component with a hole", L, 0.4, [(1, 6, 1, 6)])
    # 4. two pixels touching at a corner, (2, 2) and (3, 3): ONE 8-connected contour, rect (2, 2, 2, 2) -> the 2 x 2 square,
    #    including the two off-diagonal pixels that are off (4-connectivity would give two 1 x 1 boxes instead)
    L = _blank(0); L[2, 2] = 255; L[3, 3] = 255
    add("diagonal neighbours are one contour", L, 0.4, [(2, 4, 2, 4)])
    # 5. two separate components (a gap of one pixel column AND row between them): two boxes
    L = _blank(0); L[1:3, 1:3] = 255; L[4:6, 4:7] = 255
    add("two components", L, 0.4, [(1, 3, 1, 3), (4, 6, 4, 7)])
    # 6. all-zero map: max = 0, thresh = 0, nothing is > 0 -> no contour -> [[0,0,0,0]] -> mask[0:0, 0:0]: empty
    add("empty map", _blank(0), 0.4, [])
    # 7. strict threshold at exactly int(thr * max): max level 200, thr 0.4 -> int(80.0) = 80: level 80 is off, level 81 on
    L = _blank(0); L[0, 0] = 200; L[2, 2:5] = 80; L[4, 2:5] = 81
    add("threshold is strict at int(0.4 * 200) = 80", L, 0.4, [(0, 1, 0, 1), (4, 5, 2, 5)])
    # 8. COCO's threshold 0.7 at max 255: 0.7 * 255 = 178.49999999999997 -> int -> 178: level 178 off, 179 on
    L = _blank(0); L[0, 0] = 255; L[3, 1:4] = 178; L[5, 1:4] = 179
    add("int(0.7 * 255) = 178", L, 0.7, [(0, 1, 0, 1), (5, 6, 1, 4)])
    # 9. constant map: everything is > int(0.4 * 255) = 102: one component, rect (0, 0, 10, 8) -> box [0, 0, 9, 7]:
    #    the last row and the last column stay unmasked
    add("full map", _blank(255), 0.4, [(0, H - 1, 0, W - 1)])
    # 10. a component that lies only in the last column (rows 2..4 of column 9): rect (9, 2, 1, 3) -> x1 = min(10, 9) = 9
    #     -> mask[2:5, 9:9]: nothing; together with a corner pixel (7, 9): rect (9, 7, 1, 1) -> nothing either
    L = _blank(0); L[2:5, 9] = 255; L[7, 9] = 255
    add("last column only", L, 0.4, [])
    # 11. an L-shaped component: the rectangle covers the pixels the L leaves out
    L = _blank(0); L[1:6, 2] = 255; L[5, 2:7] = 255
    add("L shape", L, 0.4, [(1, 6, 2, 7)])
    return out
