#!/bin/bash
# Round 5's closing evidence on the committed tree: GPU tests, smoke(), bench lines in the default and the driver's form,
# kernel-trace statistics of the env-only bench.  usage (on the GPU box): tools/prof_round5_final.sh <tag>
R=$GRAFT_REPO_ROOT; tag=$1; O=$R/gpurun_out
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/${tag}_pytest.log 2>&1 || { tail -15 $O/${tag}_pytest.log; exit 1; }
tail -1 $O/${tag}_pytest.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${tag}_smoke.txt 2>&1 || { tail -5 $O/${tag}_smoke.txt; exit 1; }
tail -1 $O/${tag}_smoke.txt
timeout -k 10 400 python3 bench.py > $O/${tag}_bench_default.json 2> $O/${tag}_bench_default.err || { tail -5 $O/${tag}_bench_default.err; exit 1; }
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_bench_steps20.json 2> $O/${tag}_bench_steps20.err || exit 1
echo "bench done"
bash tools/prof_bench.sh $tag > $O/${tag}_prof_bench.txt 2>&1 || exit 1
echo "prof_bench done"
