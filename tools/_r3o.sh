O=gpurun_out/r3o; mkdir -p $O
export SURFD_BENCH_BACKEND=gloo
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 1 --no-trace --no-e2 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "n2 rc=$?"; tail -4 $O/bench_n2_gloo.err | cut -c1-300; python -c "
import json; d=json.load(open('$O/bench_n2_gloo.json')); print(d['value'], d['n_gpus'], d['rccl_ranks'], d.get('not_a_scaling_measurement'), d['per_rank'], d['config']['host_threads_per_rank'])"
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 0 --mode grid-shard --batch 2 --diffusion-steps 50 --resolution 256 > $O/bench_gs2_gloo.json 2> $O/bench_gs2_gloo.err; echo "gs2 rc=$?"; tail -4 $O/bench_gs2_gloo.err | cut -c1-300; cut -c1-700 $O/bench_gs2_gloo.json
unset SURFD_BENCH_BACKEND
timeout 300 python bench.py --steps 1 --warmup 0 --mode grid-shard --batch 2 --diffusion-steps 50 --resolution 256 2>/dev/null | cut -c1-200
