O=gpurun_out/r3r; mkdir -p $O
python tools/dec_time.py 22 2>/dev/null | tee -a $O/xbar.txt
SURFD_DECODER_XBAR=0 python tools/dec_time.py 22 2>/dev/null | tee -a $O/xbar.txt
python tools/dec_time.py 22 2>/dev/null | tee -a $O/xbar.txt
SURFD_DECODER_XBAR=0 python tools/dec_time.py 22 2>/dev/null | tee -a $O/xbar.txt
SURFD_DECODER_XBAR=20000 python tools/dec_time.py 22 2>/dev/null | tee -a $O/xbar.txt
python tools/dec_cu_sweep.py 2>/dev/null | tee -a $O/xbar.txt
timeout 600 python -m pytest tests/test_gpu_decoder_grid.py -x -q -m gpu -k "decoder_vs_golden or ragged or grid_native or batched or band" 2>&1 | tail -3
