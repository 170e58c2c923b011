"""Inference driver with the reference's ``Darknetv3Detector`` interface (reference: detect.py:22-252) on the
HIP path, with real batching.

Same constructor arguments and call protocol (``Darknetv3Detector(images, destination, cfg_path, weights_path,
resolution, confidence, nms_thresh, CUDA, TORCH)()``), same ``metrics.json`` schema
(``{image_name: [[img_idx, x1, y1, x2, y2, objectness, class_score, class], ...] | 0}``, boxes in letterbox
coordinates like the reference stores them) and ``det_<cfg>_<image>`` outputs.  Differences, all deliberate
(SURVEY.md §3.1): frames really run in batches (``batch_size`` keyword, the reference hard-codes 1 and drops
nothing only because of that), a trailing partial batch is processed, the box rescale uses the actual
resolution instead of the constant 416, the model runs in ``.eval()`` mode, images are read and
drawn with PIL (OpenCV is not a dependency), preprocessing runs on the GPU (``util.prep_image``).
"""
import json
import os
import os.path as osp
import time

import numpy as np
import torch

from .darknet import Darknet
from .util import load_classes, prep_image, rescale_boxes, write_results


class Darknetv3Detector:
    def __init__(self, images: str, destination: str, cfg_path: str, weights_path: str, resolution: int,
                 confidence: float, nms_thresh: float, CUDA: bool = True, TORCH: bool = False, batch_size: int = 8,
                 names_path: str = None, draw: bool = True):
        self.images = images
        self.batch_size = int(batch_size)
        self.confidence = float(confidence)
        self.nms_thresh = float(nms_thresh)
        self.destination = destination
        self.cfg_path = cfg_path
        self.weights_path = weights_path
        self.reso = int(resolution)
        self.CUDA = CUDA
        self.TORCH = TORCH
        self.draw = draw
        self.metrics = {}
        self.timings = {}
        self.num_classes = 80
        if names_path is None and osp.exists("data/coco.names"):
            names_path = "data/coco.names"
        self.classes = load_classes(names_path)

    # ------------------------------------------------------------------------------------------
    def configure_darknet(self):
        model = Darknet(self.cfg_path, self.CUDA)
        if self.TORCH:
            model.load_state_dict(torch.load(self.weights_path, map_location="cpu", weights_only=True))
        else:
            model.load_weights(self.weights_path)
        return model.eval()

    @staticmethod
    def read_directory(directory):
        if osp.isdir(directory):
            names = sorted(os.listdir(directory))
            return [osp.join(osp.realpath("."), directory, n) for n in names], names
        if osp.isfile(directory):
            return [osp.join(osp.realpath("."), directory)], [osp.basename(directory)]
        print("No file or directory with the name {}".format(directory))
        raise FileNotFoundError(directory)

    @staticmethod
    def _load_rgb(path):
        from PIL import Image
        with Image.open(path) as im:
            return np.asarray(im.convert("RGB"), dtype=np.uint8)

    def __call__(self, *args, **kwargs):
        model = self.configure_darknet()
        os.makedirs(self.destination, exist_ok=True)
        model.net_info["height"] = self.reso
        self.inp_dim = int(model.net_info["height"])
        assert self.inp_dim % 32 == 0
        assert self.inp_dim > 32
        paths, names = self.read_directory(self.images)
        print("Number of Images= ", len(paths))
        for start_idx in range(0, len(paths), self.batch_size):
            b_paths = paths[start_idx:start_idx + self.batch_size]
            b_names = names[start_idx:start_idx + self.batch_size]
            loaded = [self._load_rgb(p) for p in b_paths]
            im_dims = torch.tensor([(im.shape[1], im.shape[0]) for im in loaded], dtype=torch.float32)
            t0 = time.time()
            x = torch.cat([prep_image(im, self.inp_dim, mode="RGB") for im in loaded], 0)
            with torch.no_grad():
                prediction = write_results(model(x), self.num_classes, self.confidence, self.nms_thresh)
            torch.cuda.synchronize()
            dt = time.time() - t0
            for j, name in enumerate(b_names):
                if isinstance(prediction, int):
                    rows = None
                else:
                    rows = prediction[prediction[:, 0] == j].clone()
                    rows[:, 0] += start_idx                      # image index over the whole run (detect.py:101-102)
                objs = [] if rows is None else [self.classes[int(r[-1])] for r in rows]
                print("{0:20s} predicted in {1:6.3f} seconds".format(name, dt / len(b_names)))
                print("{0:20s} {1:s}".format("Objects Detected:", " ".join(objs)))
                print("----------------o----------------")
                self.metrics[name] = 0 if rows is None or rows.size(0) == 0 else rows.tolist()
                self.timings[name] = dt / len(b_names)
            if self.draw and not isinstance(prediction, int):
                boxes = rescale_boxes(prediction, im_dims, self.inp_dim).cpu()
                for j, (name, im) in enumerate(zip(b_names, loaded)):
                    self._draw_and_save(im, boxes[boxes[:, 0] == j], name)
            elif self.draw:
                for name, im in zip(b_names, loaded):
                    self._draw_and_save(im, torch.zeros((0, 8)), name)
        self.save_detection_metrics()
        return self.metrics

    def _draw_and_save(self, im, rows, name):
        from PIL import Image, ImageDraw
        img = Image.fromarray(im)
        d = ImageDraw.Draw(img)
        for r in rows:
            x1, y1, x2, y2 = [int(v) for v in r[1:5]]
            cls = int(r[-1])
            colour = ((37 * cls + 60) % 256, (91 * cls + 120) % 256, (53 * cls + 200) % 256)
            d.rectangle([x1, y1, max(x2, x1), max(y2, y1)], outline=colour, width=1)
            d.text((x1 + 2, y1 + 2), "{0} {1:.4}".format(self.classes[cls], float(r[-2])), fill=colour)
        out = "{}/det_{}_{}".format(self.destination, osp.basename(self.cfg_path)[:-4], name)
        img.save(out)

    def save_detection_metrics(self):
        with open(osp.join(self.destination, "metrics.json"), "w") as f:
            json.dump(self.metrics, f)
