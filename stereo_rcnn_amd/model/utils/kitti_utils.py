"""Host-side KITTI helpers on the inference path - reference lib/model/utils/kitti_utils.py:
`read_obj_calibration` (:97-159), `infer_boundary` (:398-437), `write_detection_results` (:440-460).
Pure host I/O / tiny numpy loops in the reference as well; label/lidar/visualisation helpers of that
file are training / viz code and out of scope."""
import math
import os

import numpy as np


class FrameCalibrationData(object):
    """Calibration of one frame (P0..P3, rectification, velodyne->cam), as in kitti_utils.py:37-66."""

    def __init__(self):
        self.p0 = self.p1 = self.p2 = self.p3 = None
        self.p2_2 = self.p2_3 = None
        self.r0_rect = None
        self.t_cam2_cam0 = None
        self.tr_velodyne_to_cam0 = None


def _row(line):
    return [float(v) for v in line.strip().split(' ')[1:] if v != '']


def read_obj_calibration(calib_path):
    """Parse a KITTI object calibration file (kitti_utils.py:97-159)."""
    with open(calib_path, 'r') as fh:
        lines = [ln for ln in fh.read().split('\n')]
    c = FrameCalibrationData()
    p = [np.reshape(_row(lines[i]), (3, 4)) for i in range(4)]
    c.p0, c.p1, c.p2, c.p3 = p
    c.p2_2 = np.copy(p[2])
    c.p2_2[0, 3] -= c.p2[0, 3]
    c.p2_3 = np.copy(p[3])
    c.p2_3[0, 3] -= c.p2[0, 3]
    c.t_cam2_cam0 = np.zeros(3)
    c.t_cam2_cam0[0] = (c.p2[0, 3] - c.p0[0, 3]) / c.p2[0, 0]
    c.r0_rect = np.reshape(_row(lines[4]), (3, 3))
    c.tr_velodyne_to_cam0 = np.reshape(_row(lines[5]), (3, 4))
    return c


def infer_boundary(im_shape, boxes_left):
    """Occlusion border of every object from the 2-D boxes alone (kitti_utils.py:398-437):
    a 1-D 'depth line' over image columns (depth ~ 1050 / y2), then per box the visible [left, right]."""
    boxes_left = np.asarray(boxes_left)
    n = boxes_left.shape[0]
    left_right = np.zeros((n, 2), dtype=np.float32)
    depth_line = np.zeros(im_shape[1] + 1, dtype=float)
    for i in range(n):
        depth = 1050.0 / boxes_left[i, 3]
        for col in range(int(boxes_left[i, 0]), int(boxes_left[i, 2]) + 1):
            pixel = depth_line[col]
            if pixel == 0.0:
                depth_line[col] = depth
            elif depth < depth_line[col]:
                depth_line[col] = (depth + pixel) / 2.0
    for i in range(n):
        d = 1050.0 / boxes_left[i, 3]
        x1, x2 = int(boxes_left[i, 0]), int(boxes_left[i, 2])
        left_right[i, 0], left_right[i, 1] = boxes_left[i, 0], boxes_left[i, 2]
        left_visible = not (depth_line[x1] < d)
        right_visible = not (depth_line[x2] < d)
        if not right_visible and not left_visible:
            left_right[i, 1] = boxes_left[i, 0]
        for col in range(x1, x2 + 1):
            if left_visible and depth_line[col] >= d:
                left_right[i, 1] = col
            elif right_visible and depth_line[col] < d:
                left_right[i, 0] = col
    return left_right


def write_detection_results(result_dir, file_number, calib, box_left, pos, dim, orien, score):
    """Append one detection to `<result_dir>/data/<file_number>.txt` in the KITTI result format consumed by
    the external evaluator (kitti_utils.py:440-460): `Car -1 -1 alpha x1 y1 x2 y2 h w l x y z ry score`,
    x moved from the cam2 to the cam0 frame, yaw reported as `orien - 1.57`."""
    if result_dir is None:
        return
    result_dir = result_dir + '/data'
    dis_cam02 = calib.t_cam2_cam0[0]
    alpha = orien - math.pi / 2 + math.atan2(-pos[0], pos[2])
    line = 'Car -1 -1 '
    line += '%f %f %f %f %f ' % (alpha, box_left[0], box_left[1], box_left[2], box_left[3])
    line += '%f %f %f %f %f %f %f %f \n' % (dim[1], dim[0], dim[2], pos[0] - dis_cam02, pos[1], pos[2],
                                            orien - 1.57, score)
    os.makedirs(result_dir, exist_ok=True)
    with open(os.path.join(result_dir, file_number + '.txt'), 'a') as fh:
        fh.write(line)
