O=gpurun_out/r3s; mkdir -p $O
B="--no-trace --no-e2 --no-cpu-baseline --timeline"
run() { tag=$1; shift; timeout 500 python bench.py --gpus 1 --steps 20 --warmup 3 $B "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python -c "
import json; d=json.load(open('$O/b_$tag.json')); print('$tag', round(d['value'],3), round(d['ms_per_step'],1), round(d['roofline']['achieved'],1), d['time_share'])"; grep timeline $O/b_$tag.err | cut -c1-160; }
run ts
run ov192_f4_lb4 --overlap-blocks 192 --first-round 4 --loop-batches 4
run ov208_f4_lb4 --overlap-blocks 208 --first-round 4 --loop-batches 4
run ov192_f2_lb3 --overlap-blocks 192 --first-round 2 --loop-batches 3
run ov176_f4_lb8 --overlap-blocks 176 --first-round 4 --loop-batches 8
