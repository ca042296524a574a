import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surfd_amd import synth
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
args = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine", sigma_small=True, clip_value=1.0)
model, _ = create_model_and_diffusion(args)
load_model_wo_clip(model, synth.synth_unet_state_dict()); model.to("cuda"); model.eval()
x = torch.randn(8, 1, 32, device="cuda"); t = torch.full((8,), 500, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    model(x, t, y={})
torch.cuda.synchronize()
