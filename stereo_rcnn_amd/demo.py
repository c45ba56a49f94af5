"""One stereo pair from disk - the reference's demo.py:63-338 without the OpenCV visualisation:

    python -m stereo_rcnn_amd.demo --left demo/left.png --right demo/right.png --calib demo/calib.txt --checkpoint models_stereo/stereo_rcnn_12_6477.pth

Prints one KITTI-format line per aligned object (class, alpha, 2-D box, h w l, x y z, ry, score) and the two phase
times the reference reports (`det_time`: forward + decode, `solve_time`: 3-D solve + dense alignment)."""
import argparse
import sys
import tempfile
import time

import torch

from . import pipeline
from .model.stereo_rcnn.resnet import resnet
from .model.utils import kitti_utils
from .test_net import read_png_rgb


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--left', required=True)
    ap.add_argument('--right', required=True)
    ap.add_argument('--calib', required=True)
    ap.add_argument('--checkpoint', required=True)
    ap.add_argument('--precision', choices=['f16x3', 'f32'], default='f16x3')
    args = ap.parse_args(argv)
    from . import serving
    serving.before_hip()
    serving.enter(1)                      # one pair: latency regime (branches on side streams, in-situ plans)
    dev = torch.device('cuda:0')
    model = resnet(('__background__', 'Car'), 101, pretrained=False)
    model.create_architecture()
    sd = torch.load(args.checkpoint, map_location='cpu')
    model.load_state_dict(sd['model'] if 'model' in sd else sd)
    model.cuda()
    model.eval()
    model.precision = args.precision
    left, right = read_png_rgb(args.left), read_png_rgb(args.right)
    calib = kitti_utils.read_obj_calibration(args.calib)
    lu, ru = torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev)
    pipeline.detect_3d_images(model, lu, ru, calib)                             # first call: per-shape autotuning
    torch.cuda.synchronize()
    t0 = time.time()
    objs = pipeline.detect_3d_images(model, lu, ru, calib)                      # preprocessing .. rectified 3-D boxes, on the device
    torch.cuda.synchronize()
    dt = time.time() - t0
    with tempfile.TemporaryDirectory() as td:
        pipeline.write_kitti_results(td, 'demo', calib, [o for o in objs if o['aligned']])
        try:
            sys.stdout.write(open(td + '/data/demo.txt').read())
        except FileNotFoundError:
            pass
    print('%d objects (%d aligned) in %.1f ms' % (len(objs), sum(o['aligned'] for o in objs), dt * 1e3))


if __name__ == '__main__':
    main()
