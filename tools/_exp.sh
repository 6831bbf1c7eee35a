timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
# 2 ranks on the one GPU over gloo: exercises the N>1 path of bench.py (result gather) after the ABI change
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --queries 20000 --backend gloo 2>&1 | tail -2 | cut -c1-400
