"""Pins the detection driver's output schema to the fixture the reference itself holds: det/metrics.json, written by
Darknetv3Detector.save_detection_metrics (detect.py:104-107) from rows collected at detect.py:155 / 164 (yolov3.cfg at 416x416,
confidence 0.6 inferred from the rows — SURVEY.md F6; it cannot be regenerated offline: no weights, no cv2).
Build container only (reads /root/reference as DATA; nothing is imported or executed):
    python tests/golden/make_metrics_schema.py
writes tests/golden/metrics_schema.json = the reference's rows (11 images, 32 rows of 8 floats, one image without detections)
plus the structural facts tests/metrics_schema.py checks a driver's metrics.json against."""
import json, os
SRC = "/root/reference/det/metrics.json"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "metrics_schema.json")
ref = json.load(open(SRC))
names = list(ref)                               # insertion order of the reference's dict = its processing order (os.listdir)
rows = [r for v in ref.values() if v != 0 for r in v]
schema = {
    "source": "uguryagmur/RealTimeObjectDetection det/metrics.json (detect.py:104-107, 155, 164)",
    "row": ["image index in processing order (run-global, float)", "x1", "y1", "x2", "y2", "objectness", "class score", "class index (float, integral)"],
    "row_len": 8,
    "no_detection_value": 0,
    "n_images": len(ref),
    "n_rows": len(rows),
    "processing_order": names,
    "min_objectness": min(r[5] for r in rows),
    "max_coord": max(max(r[1:5]) for r in rows),
    "reference_metrics": ref,
}
json.dump(schema, open(OUT, "w"), indent=1)
print(OUT, schema["n_images"], "images", schema["n_rows"], "rows")
