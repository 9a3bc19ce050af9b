#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dist; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_distributed.py -m gpu -x -q 2>&1 | tail -15
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err; tail -c 1500 $O/bench1.json
PRT_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench2.json 2> $O/bench2.err; tail -c 2500 $O/bench2.json; tail -5 $O/bench2.err
