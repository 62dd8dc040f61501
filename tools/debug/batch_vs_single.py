import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sps_amd import synthetic
from sps_amd.engine import ScanEngine
from sps_amd.models.models import SPSNet
import yaml
cfg = yaml.safe_load(open("config/config.yaml"))
torch.manual_seed(0)
net = SPSNet(cfg).cuda().eval().freeze()
scans = list(synthetic.make_sequence(8, voxel_size=0.1))
print([len(s) for s in scans])
eng = ScanEngine(net, 0, streams=3, table_rows=16)
a = eng.run_sequence([torch.from_numpy(s).pin_memory() for s in scans])
b = eng.run_sequence([torch.from_numpy(s).cuda() for s in scans])
eng.reset_table(8)
for i in (0, 4):
    eng.submit(torch.from_numpy(synthetic.collate(scans[i:i+4])), 4)
c = eng.finish().cpu().numpy()
eng1 = ScanEngine(net, 0, streams=1, table_rows=16)
d = eng1.run_sequence([torch.from_numpy(s).cuda() for s in scans])
np.set_printoptions(linewidth=200, precision=6, suppress=True)
print("pinned\n", a[:, [0,3,5,6]]); print("device\n", b[:, [0,3,5,6]]); print("batch4\n", c[:, [0,3,5,6]]); print("serial\n", d[:, [0,3,5,6]])
