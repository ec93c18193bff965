import sys, collections
sys.path.insert(0, "/root/repo")
import torch
from indm_amd import configs, hip, losses, sde_lib, utils
from indm_amd.flow import create_flow_model
from indm_amd.rng import DeviceRng
from bench import synthetic_batch
dev = torch.device("cuda")
cfg = configs.get_config("vp/CIFAR10/indm_nll"); cfg.device = dev; cfg.training.batch_size = 16; cfg.compute_dtype = "bf16"
torch.manual_seed(0)
score = utils.create_model(cfg, dev); flow = create_flow_model(cfg).to(dev)
state, fstate, _ = utils.build_states(cfg, score, flow, dev)
step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True)
score.train(); flow.train()
rng = DeviceRng(dev, seed=1, poisson_seed=2)
cnt = collections.Counter()
orig = hip.pack_kpad_bf16
def counted(ws, kind):
    cnt[(kind, len(ws))] += 1
    return orig(ws, kind)
hip.pack_kpad_bf16 = counted
orig2 = hip.pack_w2t_bf16
def counted2(ws):
    cnt[("w2t", len(ws) if ws else 0)] += 1
    return orig2(ws)
hip.pack_w2t_bf16 = counted2
for i in range(3):
    cnt.clear()
    step_fn(state, fstate, synthetic_batch(16, 32, i, dev), rng=rng)
    torch.cuda.synchronize()
    print("step", i, dict(cnt))
