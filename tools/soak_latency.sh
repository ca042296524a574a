#!/bin/bash
# soak of the latency form after the round-6 changes of its tail: many evaluations of one input at several batch sizes (the K split
# follows the batch: 1, 2, 3, 5, 8, 16 latents exercise different split-K / k-part / ragged-chunk combinations), every one compared
# bit for bit with the first; then the same under load from a second stream running wide loops (workgroups of both share CUs)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/soak; mkdir -p $O; : > $O/soak.txt
for B in 1 2 3 5 8 16; do timeout 600 python tools/determinism_check.py 300 $B 0 2>&1 | grep -E "distinct|differs" >> $O/soak.txt; done
timeout 900 python - >> $O/soak.txt 2>&1 <<'PY'
import hashlib, os, sys, threading, types
sys.path.insert(0, os.getcwd())
import torch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
from surfd_amd.diffusion import create_gaussian_diffusion
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, _ = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
wide = model.replica(); wide.set_wide(80)
diff = create_gaussian_diffusion(args, "ddim100")
stop = False
def load():
    st = torch.cuda.Stream()
    noise = synth.synth_noise_batch(diff.num_timesteps, 0, 80, 32).cuda()
    with torch.cuda.stream(st):
        while not stop:
            diff.p_sample_loop(wide, (80, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True)
            st.synchronize()
th = threading.Thread(target=load); th.start()
for B, L in ((8, 32), (5, 32), (8, 64)):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 1, L, generator=g).cuda(); t = torch.full((B,), 400, device="cuda")
    seen = {}
    for i in range(200):
        out = model(x, t, y={}); torch.cuda.synchronize()
        h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]; seen[h] = seen.get(h, 0) + 1
    print(f"under load (a wide loop of 80 on a second stream): B={B} L={L} latency form: {len(seen)} distinct output(s) in 200 runs {list(seen.values())}")
stop = True; th.join()
PY
cat $O/soak.txt
