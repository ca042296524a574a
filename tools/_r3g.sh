O=gpurun_out/r3g; mkdir -p $O
python tools/dec_time.py 22 > $O/dec_variants.txt 2>/dev/null
SURFD_LIB=$PWD/surfd_amd/lib/variants/libsurfd_hip_st.so python tools/dec_time.py 22 >> $O/dec_variants.txt 2>/dev/null
cat $O/dec_variants.txt
timeout 900 python -m pytest tests/test_gpu_decoder_grid.py -x -q -m gpu > $O/pytest_dec.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_dec.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "rc=$?"; python -c "
import json; d=json.load(open('$O/bench_driver.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['time_share'], d['breakdown_ms_per_step'], d['w_trace'])"
