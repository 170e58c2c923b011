# one gpurun call: GPU test suite, smoke(), the full profile round (bench lines, kernel stats, HBM traffic, SQ counters)
tag=${1:-r03}
o=gpurun_out/final_$tag; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1 || { tail -30 $o/pytest.log; exit 1; }
tail -3 $o/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1 || { tail -20 $o/smoke.log; exit 1; }
tail -1 $o/smoke.log
bash tools/profile_round.sh $tag || { tail -20 gpurun_out/prof_$tag/bench.err; exit 1; }
