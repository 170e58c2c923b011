#!/usr/bin/env python
"""A/B/n of several BUILDS of librtod in ONE process: every library is dlopen'ed privately (own globals, own code objects), one
Darknet per library, the same tile table installed in each (RTOD_TILES, saved by bench.py --tiles; without it each library autotunes),
per-launch hipEvent timings taken in interleaved rounds — box, clock state and thermal drift are shared, which separate processes
(tools/run_abn.sh: +-1...2 % between identical kernels) cannot offer.
    python tools/exp_ab_libs.py out.json res batch libA.so libB.so ...     (names relative to realtimeobjectdetection_amd/)"""
import collections, json, os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from realtimeobjectdetection_amd import cfgs, synth, _ffi
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet
out = sys.argv[1]; res = int(sys.argv[2]); B = int(sys.argv[3]); names = sys.argv[4:]
ROUNDS = int(os.environ.get("ROUNDS", "24"))
here = os.path.dirname(os.path.abspath(_ffi.__file__))
handles = {}
for n in names:
    _ffi.LIB_PATH = os.path.join(here, n); _ffi._lib = None
    handles[n] = _ffi.lib()
def use(n): _ffi._lib = handles[n]
text = cfgs.yolov3_cfg(); ir = build_ir(parse_cfg_text(text), res)
w = synth.synth_weights(ir)
x = torch.from_numpy(synth.synth_frames(B, res)).cuda()
d = tempfile.mkdtemp()
table = None
if os.environ.get("RTOD_TILES"):
    table = json.load(open(os.environ["RTOD_TILES"])).get("f16s3_%d_b%d" % (res, B))
models = {}
ref = None
for n in names:
    use(n)
    m = Darknet(cfgs.write_cfg(os.path.join(d, "m.cfg"), text), True).eval()
    m.net_info["height"] = res; m.precision = "f16s3"; m.overflow_check = "off"
    m.load_weight_stream(w)
    if table is not None:
        m.prepare(B); m.set_tiles(B, table)
    with torch.no_grad():
        m(x); y = m(x)
    torch.cuda.synchronize()
    if ref is None: ref = y.clone()
    elif not torch.equal(y, ref): print("!! output of", n, "differs from", names[0], "max abs", float((y - ref).abs().max()))
    models[n] = m
def cls(li, nm):
    if li.kind != 0: return "other"
    if li.ksize == 1: return "1x1"
    if li.stride == 2: return "3x3s2"
    if "band" in nm: return "band%d" % li.hout
    return "3x3s1gen"
samples = {n: [] for n in names}
with torch.no_grad():
    for r in range(ROUNDS):
        order = names if r % 2 == 0 else names[::-1]          # alternate the order: neither library always runs first
        for n in order:
            use(n)
            _, ms = models[n].forward_timed(x)
            _, ms2 = models[n].forward_timed(x)
            samples[n].append((ms + ms2) / 2)
tabs = {}; sds = {}
for n in names:
    use(n)
    st = np.stack([np.asarray(s.cpu() if hasattr(s, "cpu") else s, dtype=np.float64) for s in samples[n]])      # [rounds, launches]
    g = collections.defaultdict(lambda: np.zeros(st.shape[0]))
    for k, li in enumerate(models[n].launch_infos()):
        nm = _ffi.lib().rtod_conv_variant_name(li.variant).decode() if li.kind == 0 else ""
        g[cls(li, nm)] += st[:, k]
    g["TOTAL"] = st.sum(axis=1)
    # median and its standard error from the MAD (1.4826 MAD = sigma; x 1.2533 / sqrt(n) for the median): single slow rounds do not move it
    tabs[n] = {k: float(np.median(v)) for k, v in g.items()}
    sds[n] = {k: float(1.4826 * np.median(np.abs(v - np.median(v))) * 1.2533 / np.sqrt(len(v))) for k, v in g.items()}
keys = sorted(tabs[names[0]])
print("%-10s" % "", " ".join("%22s" % n[:22] for n in names), "   (ms per forward, median +- its standard error over %d interleaved rounds)" % ROUNDS)
for k in keys:
    print("%-10s" % k, " ".join("%13.4f +- %.4f" % (tabs[n][k], sds[n][k]) for n in names),
          " ".join("%+6.2f%%" % ((tabs[n][k] / tabs[names[0]][k] - 1) * 100) for n in names[1:]))
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
json.dump({"median": tabs, "sem": sds, "rounds": ROUNDS}, open(out, "w"), indent=1)
