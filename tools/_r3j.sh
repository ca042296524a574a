O=gpurun_out/r3j; mkdir -p $O
for v in stamps_ovl stamps_novl; do echo "== $v" >> $O/stamps.txt; SURFD_LIB=$PWD/surfd_amd/lib/variants/libsurfd_hip_$v.so python tools/dec_stamps.py >> $O/stamps.txt 2>&1; done
cat $O/stamps.txt | grep -v amdgpu.ids
