"""Deterministic synthetic weights / frames / predictions (SURVEY.md §8 d).

There is no network, so no trained ``.weights`` file exists anywhere in the pipeline.  Every
test, fixture and benchmark regenerates its inputs from the seeds below on both sides of a
comparison (oracle here, HIP path on the GPU box); only small expected-output fixtures are
committed.  numpy's PCG64 ``Generator`` streams are stable for a given numpy version and the
container image is identical on both sides.

Weight stream layout = Darknet ``.weights`` payload (reference: src/darknet.py:316-410):
per ``convolutional`` block in cfg order ``[bn.bias, bn.weight, running_mean, running_var]``
(or ``[conv.bias]``), then ``conv.weight`` in OIHW order.
"""
import numpy as np

from .cfg import NetIR

WEIGHT_SEED = 4321
FRAME_SEED = 1234
PRED_SEED = 2024

# Head (linear 1x1, Cout = A*(5+C)) statistics: raw logit std and objectness bias chosen so that
# roughly 1-2 % of anchors exceed conf 0.6 (the load SURVEY.md §6 measured on the reference).
HEAD_LOGIT_STD = 2.0
HEAD_OBJ_BIAS = -4.0
# Per-(head, anchor) objectness biases for WEIGHT_SEED, measured by tests/golden/calibrate_heads.py
# so that ~1.4 % of anchors exceed conf 0.6 on synth_frames (with random weights on noise frames an
# objectness channel is a per-channel constant plus small spatial noise, so one flat bias gives
# either no candidates or most of the grid).
HEAD_OBJ_BIAS_TABLE = {
    "yolov3-tiny": [[-0.6145, 0.6579, -2.1158], [0.5275, -2.0477, 0.4183]],
    "yolov3": [[-1.2112, -3.8612, -0.0726], [-0.1438, -1.3899, -5.3231], [0.7336, -0.437, -4.7752]],
}


def _second_moments(ir: NetIR):
    """Expected E[x^2] of every layer output under ``synth_weights`` (analytic, from the IR).

    Random BN statistics do not normalise anything, so a naive He init grows activations by
    ~1.6x per layer (and 2x per shortcut) and overflows after 75 layers.  Instead each conv's
    weight std is chosen from the expected second moment of its input so that its output has
    E[x^2] ~ 1; this function propagates those expectations through route/shortcut/pool layers.
    Returns (m2_in per conv layer index, m2_out per layer).
    """
    m2_out, mean_out, m2_in = [], [], {}
    prev, prev_mean = 1.0 / 3.0, 0.5                   # x ~ U[0,1)
    # post-leaky mean of z ~ N(0, s^2): 0.9 * s / sqrt(2 pi); with E[leaky(z)^2] = 1 -> s^2 = 1/0.505
    leaky_mean = 0.9 / np.sqrt(2 * np.pi) / np.sqrt(0.505)
    for L in ir.layers:
        if L.type == "convolutional":
            m2_in[L.index] = prev
            cur = 1.0 if (L.bn or L.leaky) else HEAD_LOGIT_STD ** 2
            cur_mean = leaky_mean if L.leaky else 0.0
        elif L.type == "shortcut":
            a, b = L.srcs
            cur = m2_out[a] + m2_out[b] + 2.0 * mean_out[a] * mean_out[b]
            cur_mean = mean_out[a] + mean_out[b]
        elif L.type == "route":
            ch = [ir.layers[s].cout for s in L.srcs]
            cur = sum(c * m2_out[s] for c, s in zip(ch, L.srcs)) / float(sum(ch))
            cur_mean = sum(c * mean_out[s] for c, s in zip(ch, L.srcs)) / float(sum(ch))
        elif L.type == "maxpool":
            cur, cur_mean = prev * 1.55, prev_mean * 1.6
        elif L.type == "yolo":
            cur, cur_mean = m2_out[L.index - 1], mean_out[L.index - 1]
        else:                                          # upsample
            cur, cur_mean = prev, prev_mean
        m2_out.append(cur)
        mean_out.append(cur_mean)
        prev, prev_mean = cur, cur_mean
    return m2_in, m2_out


# E[gamma^2] * E[1/var] for gamma ~ U[0.8,1.6], var ~ U[0.5,1.5]
_BN_GAIN = ((1.6 ** 3 - 0.8 ** 3) / (3 * 0.8)) * np.log(3.0)
_LEAKY_GAIN = 0.5 * (1.0 + 0.1 ** 2)


def _net_name(ir: NetIR) -> str:
    n = len(ir.layers)
    return {107: "yolov3", 24: "yolov3-tiny"}.get(n, "")


def synth_weights(ir: NetIR, seed: int = WEIGHT_SEED, obj_bias_table=None) -> np.ndarray:
    """float32 weight stream for ``ir`` in ``.weights`` order."""
    if obj_bias_table is None:
        obj_bias_table = HEAD_OBJ_BIAS_TABLE
    head_rows = obj_bias_table.get(_net_name(ir), [])
    head_no = 0
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty(ir.n_weights, dtype=np.float32)
    m2_in, _ = _second_moments(ir)
    p = 0
    for L in ir.layers:
        if L.type != "convolutional":
            continue
        c, k = L.cout, L.cin * L.size * L.size
        if L.bn:
            out[p:p + c] = rng.normal(0.0, 0.1, c); p += c          # beta
            out[p:p + c] = rng.uniform(0.8, 1.6, c); p += c         # gamma
            out[p:p + c] = rng.normal(0.0, 0.1, c); p += c          # running_mean
            out[p:p + c] = rng.uniform(0.5, 1.5, c); p += c         # running_var
            gain = _BN_GAIN * (_LEAKY_GAIN if L.leaky else 1.0)
            std = np.sqrt(1.0 / (k * m2_in[L.index] * gain))
        else:
            bias = np.zeros(c, dtype=np.float32)
            attrs = ir.attrs if ir.attrs and c % ir.attrs == 0 else 0
            if attrs:
                bias[4::attrs] = HEAD_OBJ_BIAS
                if seed == WEIGHT_SEED and head_no < len(head_rows) and len(head_rows[head_no]) == c // attrs:
                    bias[4::attrs] = head_rows[head_no]
                head_no += 1
            out[p:p + c] = bias; p += c
            std = HEAD_LOGIT_STD / np.sqrt(k * m2_in[L.index])
        n = c * k
        out[p:p + n] = rng.standard_normal(n, dtype=np.float32) * np.float32(std); p += n
    assert p == out.size
    return out


def write_weights_file(path: str, weights: np.ndarray, seen: int = 0) -> str:
    """Darknet binary: int32[5] header (``seen`` = header[3]) + float32 stream."""
    header = np.array([0, 2, 0, seen, 0], dtype=np.int32)
    with open(path, "wb") as f:
        header.tofile(f)
        np.ascontiguousarray(weights, dtype=np.float32).tofile(f)
    return path


def read_weights_file(path: str):
    with open(path, "rb") as f:
        header = np.fromfile(f, dtype=np.int32, count=5)
        weights = np.fromfile(f, dtype=np.float32)
    return header, weights


def synth_frames(batch: int, res: int, seed: int = FRAME_SEED) -> np.ndarray:
    """``x ~ U[0,1)`` float32 ``[B,3,R,R]`` (RGB, 0..1 like prep_image's output)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random((batch, 3, res, res), dtype=np.float32)


def synth_predictions(batch: int, n: int, classes: int, res: int, seed: int = PRED_SEED,
                      obj_mu: float = -4.0, obj_sigma: float = 2.0) -> np.ndarray:
    """Stand-alone ``write_results`` input ``[B,N,5+C]`` (decoded head output).

    ``xy ~ U[0,R)``, ``wh = 60*exp(N(0,0.4))``, ``obj = sigmoid(N(mu,sigma))``, class
    probabilities ``sigmoid(N(-2,2))`` with +6 on one of 4 "hot" classes per anchor so real
    suppression happens.  Objectness values are made pairwise distinct (torch.sort is not
    stable, so ties would leave the reference's order undefined).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    p = np.empty((batch, n, 5 + classes), dtype=np.float32)
    p[..., 0:2] = rng.uniform(0, res, (batch, n, 2))
    p[..., 2:4] = 60.0 * np.exp(rng.normal(0, 0.4, (batch, n, 2)))
    logit = rng.normal(obj_mu, obj_sigma, (batch, n))
    p[..., 4] = 1.0 / (1.0 + np.exp(-logit))
    cl = rng.normal(-2.0, 2.0, (batch, n, classes))
    hot = rng.integers(0, min(4, classes), (batch, n))
    np.put_along_axis(cl, hot[..., None], np.take_along_axis(cl, hot[..., None], 2) + 6.0, 2)
    p[..., 5:] = 1.0 / (1.0 + np.exp(-cl))
    # enforce distinct objectness per image
    for b in range(batch):
        o = p[b, :, 4]
        u, idx = np.unique(o, return_index=True)
        if u.size != o.size:
            dup = np.setdiff1d(np.arange(n), idx)
            for d in dup:
                v = o[d]
                while v in u:
                    v = np.nextafter(v, np.float32(1.0), dtype=np.float32)
                o[d] = v
                u = np.append(u, v)
    return p


def conv_weight_slices(ir: NetIR):
    """``{layer index: (start, stop)}`` of each conv's OIHW weight block inside the ``.weights`` float stream."""
    out, p = {}, 0
    for L in ir.layers:
        if L.type != "convolutional":
            continue
        p += (4 if L.bn else 1) * L.cout
        n = L.cout * L.cin * L.size * L.size
        out[L.index] = (p, p + n)
        p += n
    return out


def scale_conv_weights(ir: NetIR, w: np.ndarray, factor: float, layers=None) -> np.ndarray:
    """Copy of the stream with the weight blocks of ``layers`` (default: every linear head conv) multiplied by ``factor``:
    drives head logits to |t| ~ factor * HEAD_LOGIT_STD, or an inner layer out of the split-f16 range."""
    sl = conv_weight_slices(ir)
    if layers is None:
        layers = [L.index for L in ir.layers if L.type == "convolutional" and not L.bn]
    out = w.copy()
    for i in layers:
        a, b = sl[i]
        out[a:b] *= np.float32(factor)
    return out
