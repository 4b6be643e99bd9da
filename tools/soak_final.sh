#!/bin/bash
# in-flight soaks of the shipped library (views in flight on two-stream graphs, with and without a fourth stream of hipBLASLt GEMMs), then the
# rows-per-wave / tile rule sweep on the same box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
python tools/stress_graph.py --rounds 240 --size 192x256 --ndepths 8,8,8 2>&1 | tail -1
python tools/stress_graph.py --rounds 60 --gemm 100 --size 192x256 --ndepths 8,8,8 2>&1 | tail -1
python tools/stress_graph.py --rounds 60 --size 576x800 --ndepths 48,8,8 2>&1 | tail -1
python tools/stress_graph.py --rounds 30 --gemm 200 --size 576x800 --ndepths 48,8,8 2>&1 | tail -1
python tools/stress_graph.py --rounds 30 --size 1184x1600 --ndepths 48,8,8 2>&1 | tail -1
