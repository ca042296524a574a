O=gpurun_out/r3i; mkdir -p $O
python tools/dec_time.py 22 > $O/dec_variants.txt 2>/dev/null
SURFD_LIB=$PWD/surfd_amd/lib/variants/libsurfd_hip_novl.so python tools/dec_time.py 22 >> $O/dec_variants.txt 2>/dev/null
python tools/dec_time.py 22 >> $O/dec_variants.txt 2>/dev/null
cat $O/dec_variants.txt
timeout 900 python -m pytest tests/test_gpu_decoder_grid.py -x -q -m gpu -k "decoder or grid or mesh or sharded or batched" > $O/pytest_dec.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_dec.log
