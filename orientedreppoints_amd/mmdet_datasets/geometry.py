"""Rotated-box geometry of the data pipeline.

poly2rbox / rbox2poly / get_best_begin_point: the numpy expressions of mmdet/core/bbox/transforms.py:401-500, verbatim in
arithmetic (tests pin them to the reference's own functions).  min_area_rect / box_points stand in for cv2.minAreaRect /
cv2.boxPoints (cv2 is not available): exact rotating-calipers rectangle over the convex hull in float64, returned in
OpenCV's (centre, (width, height), angle-in-degrees) form with the >= 4.5.1 angle convention (angle in (0, 90]); for a
non-degenerate quad the CORNER SET equals OpenCV's, the corner order follows cv2.boxPoints' formula."""
import math

import numpy as np

PI = np.pi


def cal_line_length(point1, point2):
    return math.sqrt(math.pow(point1[0] - point2[0], 2) + math.pow(point1[1] - point2[1], 2))


def get_best_begin_point_single(coordinate):
    x1, y1, x2, y2, x3, y3, x4, y4 = coordinate
    xmin, ymin = min(x1, x2, x3, x4), min(y1, y2, y3, y4)
    xmax, ymax = max(x1, x2, x3, x4), max(y1, y2, y3, y4)
    pts = [[x1, y1], [x2, y2], [x3, y3], [x4, y4]]
    combinate = [pts[i:] + pts[:i] for i in range(4)]
    dst = [[xmin, ymin], [xmax, ymin], [xmax, ymax], [xmin, ymax]]
    force, force_flag = 100000000.0, 0
    for i in range(4):
        temp = sum(cal_line_length(combinate[i][k], dst[k]) for k in range(4))
        if temp < force:
            force, force_flag = temp, i
    return np.array(combinate[force_flag]).reshape(8)


def get_best_begin_point(coordinates):
    return np.array(list(map(get_best_begin_point_single, coordinates.tolist())))


def rbox2poly(rrects):
    """[x_ctr, y_ctr, w, h, angle(rad)] -> [x0,y0,...,x3,y3] float32, best begin point first (transforms.py:401-421)."""
    polys = []
    for rrect in rrects:
        x_ctr, y_ctr, width, height, angle = rrect[:5]
        tl_x, tl_y, br_x, br_y = -width / 2, -height / 2, width / 2, height / 2
        rect = np.array([[tl_x, br_x, br_x, tl_x], [tl_y, tl_y, br_y, br_y]])
        R = np.array([[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]])
        poly = R.dot(rect)
        x0, x1, x2, x3 = poly[0, :4] + x_ctr
        y0, y1, y2, y3 = poly[1, :4] + y_ctr
        polys.append(np.array([x0, y0, x1, y1, x2, y2, x3, y3], dtype=np.float32))
    polys = np.array(polys)
    return get_best_begin_point(polys)


def poly2rbox(polys):
    """[x0,y0,...,x3,y3] -> [x_ctr, y_ctr, w, h, angle(rad)] with w the longer edge, angle in [-pi/4, 3pi/4)
    (transforms.py:424-466)."""
    rrects = []
    for poly in polys:
        poly = np.array(poly[:8], dtype=np.float32)
        pt1, pt2, pt3, pt4 = (poly[0], poly[1]), (poly[2], poly[3]), (poly[4], poly[5]), (poly[6], poly[7])
        edge1 = np.sqrt((pt1[0] - pt2[0]) * (pt1[0] - pt2[0]) + (pt1[1] - pt2[1]) * (pt1[1] - pt2[1]))
        edge2 = np.sqrt((pt2[0] - pt3[0]) * (pt2[0] - pt3[0]) + (pt2[1] - pt3[1]) * (pt2[1] - pt3[1]))
        angle = width = height = 0
        if edge1 > edge2:
            width, height = edge1, edge2
            angle = np.arctan2(float(pt2[1] - pt1[1]), float(pt2[0] - pt1[0]))
        elif edge2 >= edge1:
            width, height = edge2, edge1
            angle = np.arctan2(float(pt4[1] - pt1[1]), float(pt4[0] - pt1[0]))
        angle = (angle + PI / 4) % PI - PI / 4
        x_ctr = float(pt1[0] + pt3[0]) / 2
        y_ctr = float(pt1[1] + pt3[1]) / 2
        rrects.append(np.array([x_ctr, y_ctr, width, height, angle]))
    return np.array(rrects)


def _convex_hull(pts):
    """Andrew's monotone chain, counter-clockwise, collinear points dropped."""
    p = sorted(set(map(tuple, pts)))
    if len(p) <= 2:
        return p

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lower, upper = [], []
    for q in p:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], q) <= 0:
            lower.pop()
        lower.append(q)
    for q in reversed(p):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], q) <= 0:
            upper.pop()
        upper.append(q)
    return lower[:-1] + upper[:-1]


def min_area_rect(points):
    """((cx, cy), (w, h), angle_deg) of the minimum-area enclosing rectangle of `points` [n,2] (cv2.minAreaRect role)."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    hull = _convex_hull(pts)
    if len(hull) == 0:
        return (0.0, 0.0), (0.0, 0.0), 0.0
    if len(hull) == 1:
        return (float(hull[0][0]), float(hull[0][1])), (0.0, 0.0), 90.0
    h = np.array(hull, dtype=np.float64)
    best = None
    n = len(h)
    for i in range(n if n > 2 else 1):
        e = h[(i + 1) % n] - h[i]
        L = math.hypot(e[0], e[1])
        if L == 0:
            continue
        ux, uy = e[0] / L, e[1] / L                       # edge direction, its normal = (-uy, ux)
        a = h[:, 0] * ux + h[:, 1] * uy
        b = -h[:, 0] * uy + h[:, 1] * ux
        w, hh = a.max() - a.min(), b.max() - b.min()
        if best is None or w * hh < best[0]:
            ca, cb = (a.max() + a.min()) / 2, (b.max() + b.min()) / 2
            best = (w * hh, (ca * ux - cb * uy, ca * uy + cb * ux), w, hh, math.degrees(math.atan2(uy, ux)))
    _, (cx, cy), w, hh, ang = best
    # OpenCV >= 4.5.1: angle in (0, 90], width measured along the direction at `angle`
    ang = ang % 180.0
    if ang > 90.0:
        ang -= 90.0
        w, hh = hh, w
    if ang == 0.0:
        ang = 90.0
        w, hh = hh, w
    return (float(cx), float(cy)), (float(w), float(hh)), float(ang)


def box_points(rect):
    """The four corners of ((cx, cy), (w, h), angle_deg) in cv2.boxPoints' order -> float32 [4,2]."""
    (cx, cy), (w, h), ang = rect
    a = math.radians(ang)
    b, a_ = math.cos(a) * 0.5, math.sin(a) * 0.5
    p0 = (cx - a_ * h - b * w, cy + b * h - a_ * w)
    p1 = (cx + a_ * h - b * w, cy - b * h - a_ * w)
    p2 = (2 * cx - p0[0], 2 * cy - p0[1])
    p3 = (2 * cx - p1[0], 2 * cy - p1[1])
    return np.array([p0, p1, p2, p3], dtype=np.float32)
