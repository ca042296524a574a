O=gpurun_out/r3c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "wide" > $O/pytest_wide.log 2>&1; echo "pytest rc=$?" >> $O/pytest_wide.log
tail -3 $O/pytest_wide.log
timeout 400 python tools/loop_chain_sweep.py 32 1:32,2:32,3:32,4:32,2:64,3:64,4:48 1000 32 > $O/chains_wide32.md 2> $O/chains.err
cat $O/chains_wide32.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o k -- python $R/tools/loop_batch_sweep.py 32 64 50 32 > $R/$O/prof.log 2>&1
find $R/$O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-160 {} | head -8'
find $R/$O/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $R/$O/ktrace.csv
python - <<'PY'
import csv, collections, os
R=os.environ["GRAFT_REPO_ROOT"]; p=R+"/gpurun_out/r3c/ktrace.csv"
rows=list(csv.DictReader(open(p)))
rows=[r for r in rows if "conv2" in r["Kernel_Name"] or "attn_kernel" in r["Kernel_Name"] or "loop_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last evaluation's launches: duration, gap to previous end, grid size
n=104
tail=rows[-n*3:-n*2] if len(rows)>n*3 else rows[-n:]
prev=None
out=open(R+"/gpurun_out/r3c/last_eval_launches.txt","w")
for r in tail:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev)/1e3 if prev else 0
    out.write(f'{r["Kernel_Name"][:40]:40s} grid {r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size","?"):>8s} wg {r.get("Workgroup_Size_X","?"):>4s} dur {(e-s)/1e3:7.2f} us gap {gap:6.2f} us\n')
    prev=e
out.close()
os.remove(p)
PY
head -110 $R/$O/last_eval_launches.txt
find $R/$O/prof -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $R/$O/prof -name "*.db" -delete
