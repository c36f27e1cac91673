# A/B of the eval-shaped schedule's encoder form: which levels take date-paired tiles (engine.eval_pair)
python - <<'PY'
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import torch
from fabric_amd import BiDateNet
from fabric_amd.utils import inference as inf
torch.manual_seed(0)
model = BiDateNet(13, 2, precision='bf16').cuda().eval()
g = torch.Generator(device='cuda').manual_seed(3)
h = w = 8192
g1 = torch.randn(13, h, w, device='cuda', generator=g); g2 = g1 + 0.3 * torch.randn(13, h, w, device='cuda', generator=g)
n = len(inf.tile_origins(h, w, 128)[0])
eng = model.engine()
cfgs = [(), (1, 2, 3, 4, 5), (2, 3, 4, 5), (1,), (5,), (2, 3, 4)]
for rnd in range(2):
    for cfg in cfgs:
        eng.eval_pair = cfg
        inf.predict_scene(model, g1, g2, 128, 256, two_streams=False); torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            t = time.time(); inf.predict_scene(model, g1, g2, 128, 256, two_streams=False); torch.cuda.synchronize(); best = min(best, time.time() - t)
        print(cfg, round(n / best, 1), 'tiles/s')
PY
