"""`resnet` - reference lib/model/stereo_rcnn/resnet.py:220-348: the ResNet-101 + FPN + heads
parameter layout behind `_StereoRCNN`.

The modules below exist to reproduce the reference's state_dict keys and shapes
(`RCNN_layerN.0.<blk>.convK.weight`, ...), so released checkpoints load with
`load_state_dict(checkpoint['model'])`.  They are never executed: the forward runs in the
HIP library (see plan.py).
"""
import torch.nn as nn

from ..utils.config import cfg
from .stereo_rcnn import _StereoRCNN

LAYERS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


class _BottleneckParams(nn.Module):
    """Caffe-style bottleneck: the stride sits on the first 1x1 conv (resnet.py:71-74)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, with_downsample):
        super(_BottleneckParams, self).__init__()
        out = planes * self.expansion
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, out, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(out)
        if with_downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, out, 1, stride, bias=False), nn.BatchNorm2d(out))
        self.stride = stride


def _stage(inplanes, planes, blocks, stride):
    mods = [_BottleneckParams(inplanes, planes, stride, True)]
    mods += [_BottleneckParams(planes * 4, planes, 1, False) for _ in range(1, blocks)]
    return nn.Sequential(*mods)


class resnet(_StereoRCNN):
    def __init__(self, classes, num_layers=101, pretrained=False):
        self.model_path = 'data/pretrained_model/resnet101_caffe.pth'
        self.dout_base_model = 256
        self.pretrained = pretrained
        # The reference ignores num_layers and always builds ResNet-101 (resnet.py:229).
        # Here 101 is the default and 50/152 are honoured as an extension (BASELINE config 5).
        self.num_layers = num_layers if num_layers in LAYERS else 101
        _StereoRCNN.__init__(self, classes)

    def _init_modules(self):
        if self.pretrained:
            raise NotImplementedError("ImageNet initialisation is a training concern; load a checkpoint instead")
        nb = LAYERS[self.num_layers]
        stem = [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                nn.MaxPool2d(3, 2, 0, ceil_mode=True)]
        self.RCNN_layer0 = nn.Sequential(*stem)
        self.RCNN_layer1 = nn.Sequential(_stage(64, 64, nb[0], 1))
        self.RCNN_layer2 = nn.Sequential(_stage(256, 128, nb[1], 2))
        self.RCNN_layer3 = nn.Sequential(_stage(512, 256, nb[2], 2))
        self.RCNN_layer4 = nn.Sequential(_stage(1024, 512, nb[3], 2))
        self.RCNN_toplayer = nn.Conv2d(2048, 256, 1)
        self.RCNN_smooth1 = nn.Conv2d(256, 256, 3, 1, 1)
        self.RCNN_smooth2 = nn.Conv2d(256, 256, 3, 1, 1)
        self.RCNN_smooth3 = nn.Conv2d(256, 256, 3, 1, 1)
        self.RCNN_latlayer1 = nn.Conv2d(1024, 256, 1)
        self.RCNN_latlayer2 = nn.Conv2d(512, 256, 1)
        self.RCNN_latlayer3 = nn.Conv2d(256, 256, 1)
        P = cfg.POOLING_SIZE
        self.RCNN_top = nn.Sequential(nn.Conv2d(512, 2048, P, P, 0), nn.ReLU(True), nn.Dropout(p=0.2),
                                      nn.Conv2d(2048, 2048, 1), nn.ReLU(True), nn.Dropout(p=0.2))
        tower = []
        for _ in range(6):
            tower += [nn.Conv2d(256, 256, 3, 1, 1), nn.ReLU(True)]
        tower += [nn.ConvTranspose2d(256, 256, 2, 2), nn.ReLU(True)]
        self.RCNN_kpts = nn.Sequential(*tower)
        self.RCNN_cls_score = nn.Linear(2048, self.n_classes)
        self.RCNN_bbox_pred = nn.Linear(2048, 6 * self.n_classes)
        self.RCNN_dim_orien_pred = nn.Linear(2048, 5 * self.n_classes)
        self.kpts_class = nn.Conv2d(256, 6, 1)
        for p in self.parameters():
            p.requires_grad = False
        nn.Module.train(self, False)
