set -e
o=gpurun_out/r3l; mkdir -p $o
timeout -k 10 900 python bench.py --steps 20 --warmup 5 --tiles $o/tiles.json > $o/bench.log 2> $o/bench.err || (tail -30 $o/bench.err; exit 1)
tail -c 6000 $o/bench.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --tiles $o/tiles.json --no-cpu-baseline --no-extras > $o/bench2.log 2>> $o/bench.err || (tail -30 $o/bench.err; exit 1)
python - <<PY
import json
a=json.loads([l for l in open("$o/bench.log") if l.startswith("{")][-1]); b=json.loads([l for l in open("$o/bench2.log") if l.startswith("{")][-1])
print("run1", a["value"], a["single_stream"]["value"], a["dropin_api"]["value"], a["config"]["tiles"], a["roofline"]["frac"])
print("run2", b["value"], b["single_stream"]["value"], b["dropin_api"]["value"], b["config"]["tiles"], b["roofline"]["frac"])
PY
