"""Scene leg of bench.py alone, for several band heights of the host-fed path (test infrastructure):  python tools/bench_scene_hostfed.py [rows ...]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fabric_amd.utils import inference as _inf
if os.environ.get('BAND2D') is not None:          # A/B: 1 = one 2-D copy per band and date (default), 0 = one copy per plane
    _inf._SceneFeeder.band_copy_2d = os.environ['BAND2D'] == '1'
for rows in [int(a) for a in sys.argv[1:]] or [None]:
    out = bench.scene_leg(torch.device('cuda', 0), band_rows=rows)
    hf = out['host_fed']
    print(rows, 'resident', round(out['seconds'], 4), 'fed', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in hf.items() if k != 'how'})
