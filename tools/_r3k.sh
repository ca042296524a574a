O=gpurun_out/r3k; mkdir -p $O
python tools/dec_time.py 22 > $O/dec_variants.txt 2>/dev/null
SURFD_DECODER_FWD8=1 python tools/dec_time.py 22 >> $O/dec_variants.txt 2>$O/fwd8.err
python tools/dec_time.py 22 >> $O/dec_variants.txt 2>/dev/null
SURFD_DECODER_FWD8=1 python tools/dec_time.py 22 >> $O/dec_variants.txt 2>>$O/fwd8.err
cat $O/dec_variants.txt; tail -3 $O/fwd8.err
SURFD_DECODER_FWD8=1 timeout 900 python -m pytest tests/test_gpu_decoder_grid.py -x -q -m gpu -k "decoder or grid or sharded or batched" > $O/pytest_dec8.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_dec8.log
