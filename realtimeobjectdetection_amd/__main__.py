"""``python -m realtimeobjectdetection_amd detect`` — reads ``params.json`` like the reference's main.py
(reference: main.py:7-74; same ``detector_params`` keys, tabs / trailing commas tolerated).  Only the YOLOv3
detection path exists here: ``train`` and ``yolo_version: 5`` (a remote torch.hub fetch) are out of scope."""
import json
import sys


def configure_json(json_path):
    with open(json_path, "r") as f:
        s = f.read().replace("\t", "").replace("\n", "").replace(",}", "}").replace(",]", "]")
    return json.loads(s)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not argv or argv[0] != "detect":
        raise SystemExit("usage: python -m realtimeobjectdetection_amd detect [params.json]  (training is out of scope)")
    p = configure_json(argv[1] if len(argv) > 1 else "params.json")["detector_params"]
    if int(p.get("yolo_version", 3)) != 3:
        raise SystemExit("only yolo_version 3 is implemented (version 5 is a remote torch.hub fetch in the reference)")
    from .detect import Darknetv3Detector
    det = Darknetv3Detector(images=p["images_path"], destination=p["destination_path"], cfg_path=p["cfg_file_path"],
                            weights_path=p["weights_file_path"], resolution=p["resolution"], confidence=p["confidence"],
                            nms_thresh=p["nms_threshold"], CUDA=p.get("CUDA", True), TORCH=p.get("use_torch_weights", False),
                            batch_size=p.get("batch_size", 8))
    det()


if __name__ == "__main__":
    main()
