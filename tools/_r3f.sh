O=gpurun_out/r3f; mkdir -p $O
python tools/dec_time.py 22 > $O/dec_variants.txt 2>/dev/null
for v in r2dec a3 a2e a1e a0e a2i a1i a0i; do SURFD_LIB=$PWD/surfd_amd/lib/variants/libsurfd_hip_$v.so timeout 120 python tools/dec_time.py 22 >> $O/dec_variants.txt 2>/dev/null; done
python tools/dec_time.py 22 >> $O/dec_variants.txt 2>/dev/null
cat $O/dec_variants.txt
