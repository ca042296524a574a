#!/bin/bash
# Round profile set (run on the GPU box through gpurun): the driver's bench command plain and under
# rocprofv3 --kernel-trace --stats, then two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs as
# MI355X_MICROARCH.md prescribes) on a shortened run of the same kernels.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=${PROF_DIR:-gpurun_out/r06prof}; mkdir -p $O
# PMC_ONLY=1: only the counter passes (after a kernel-source change that leaves the timings of the committed runs valid)
if [ -z "${PMC_ONLY:-}" ]; then
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-trace --no-e2 --no-strict --no-trace-e2 > $O/bench_under_rocprof.json 2> $O/kt.log
fi
SHORT="python bench.py --steps 20 --warmup 0 --diffusion-steps 20 --no-cpu-baseline --no-trace --no-e2 --no-strict --no-trace-e2"      # the driver's 20 steps (loops of 80 latents, all grids), 20 instead of 1000 loop iterations
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $SHORT > $O/pmc_fetch.json 2> $O/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $SHORT > $O/pmc_write.json 2> $O/pmc_write.log
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o s -- $SHORT > $O/pmc_sq.json 2> $O/pmc_sq.log
# keep only the summaries (the raw traces are large)
python - <<'PY'
import csv, collections, glob, json, os
O = os.environ.get("PROF_DIR", "gpurun_out/r06prof")
def agg(path, col):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:70]
        d[k][0] += 1; d[k][1] += float(r[col])
    return d
out = {}
for tag, pat in [("fetch", "pmc_fetch/**/*counter_collection.csv"), ("write", "pmc_write/**/*counter_collection.csv"), ("sq", "pmc_sq/**/*counter_collection.csv")]:
    files = glob.glob(os.path.join(O, pat), recursive=True)
    if not files: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen: seen.add(key); cnt[k] += 1
    out[tag] = {k: {"dispatches": cnt[k], **v} for k, v in acc.items()}
import hashlib
out["source_sha256"] = {rel: hashlib.sha256(open(rel, "rb").read()).hexdigest()
                        for rel in ("surfd_amd/csrc/decoder.hip", "surfd_amd/csrc/conv_f16x2.hip", "surfd_amd/csrc/unet.hip")}
json.dump(out, open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(os.path.join(O, d, "**", "*"), recursive=True):
        if os.path.isfile(f) and os.path.getsize(f) > 2_000_000: os.remove(f)
for f in glob.glob(os.path.join(O, "kt", "**", "*kernel_trace.csv"), recursive=True): os.remove(f)
PY
ls -la $O $O/kt/* 2>/dev/null | head -30
