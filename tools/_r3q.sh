O=gpurun_out/r3q; mkdir -p $O
for rep in 1 2; do
python tools/dec_time.py 22 >> $O/dec_variants.txt 2>/dev/null
SURFD_LIB=$PWD/surfd_amd/lib/variants/libsurfd_hip_pkfma.so python tools/dec_time.py 22 >> $O/dec_variants.txt 2>/dev/null
done
cat $O/dec_variants.txt
