"""Validator of a detection driver's metrics.json against the schema of the fixture the reference holds (det/metrics.json,
tests/golden/metrics_schema.json <- tests/golden/make_metrics_schema.py).  Reference: detect.py:104-107 (dump), 101-102 / 155
(run-global image index in column 0), 164 (the int 0 for an image without detections), src/util.py:332-341 (row layout)."""
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_schema.json")


def load_schema():
    return json.load(open(GOLDEN))


def validate_metrics(metrics, processing_order, num_classes, confidence, schema=None):
    """`metrics`: the parsed metrics.json; `processing_order`: image file names in the order the driver processed them.
    Raises AssertionError naming the first violated rule."""
    schema = schema or load_schema()
    assert isinstance(metrics, dict) and set(metrics) == set(processing_order), "keys are exactly the image file names"
    index_of = {n: i for i, n in enumerate(processing_order)}
    for name, v in metrics.items():
        if v == schema["no_detection_value"] and not isinstance(v, list):
            assert isinstance(v, int), "an image without detections maps to the int 0"
            continue
        assert isinstance(v, list) and len(v) > 0, "detections are a non-empty list of rows (never an empty list: that is the int 0)"
        prev = None
        for r in v:
            assert isinstance(r, list) and len(r) == schema["row_len"] and all(isinstance(x, float) for x in r), "a row is 8 floats"
            img, x1, y1, x2, y2, obj, score, cls = r
            assert img == float(index_of[name]), "column 0 is the image's run-global index in processing order"
            assert cls == float(int(cls)) and 0 <= int(cls) < num_classes, "column 7 is an integral class index"
            assert confidence < obj <= 1.0 and 0.0 < score <= 1.0, "objectness above the confidence threshold, class score in (0, 1]"
            assert x1 < x2 and y1 < y2 and all(abs(c) < float("inf") for c in (x1, y1, x2, y2)), "finite corner boxes x1 < x2, y1 < y2 (network-input pixels)"
            key = (cls, -obj)
            assert prev is None or prev <= key, "rows of an image: class ascending, objectness descending inside a class (write_results order)"
            prev = key
