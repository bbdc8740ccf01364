"""nrv_predict_reads_raw from host memory against the device-resident read-mode call on the same reads (T = 11, human weights):
where the 11 % go (NRV_HOST_TRACE=2: kernel span and gap per stage).   python3 scripts/gpu_rawpath.py [reps]"""
import os, sys, time
import numpy as np, torch
os.environ.setdefault("NRV_HOST_TRACE", "2")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
reads = bench.fixture_reads()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reads = reads * k
a, b = load_species("human")
rv = Reviser(a, b, device=0, batch=4096)
rv.set_stream(torch.cuda.current_stream().cuda_stream)
raws = [[r.raw for r in reads], [r.starts for r in reads], [r.feat_ev for r in reads], [r.shift for r in reads], [r.scale for r in reads]]
sev = np.concatenate([r.sig_ev for r in reads]); fev = np.concatenate([r.feat_ev for r in reads])
n = len(fev) - a.T
dev = "cuda:0"
d_sev, d_fev = torch.from_numpy(sev).to(dev), torch.from_numpy(fev).to(dev)
o = (torch.empty(n, 6, device=dev), torch.empty(n, 5, device=dev), torch.empty(n, dtype=torch.int8, device=dev), torch.empty(n, dtype=torch.int8, device=dev))
rp = (d_sev.data_ptr(), d_fev.data_ptr(), len(fev)) + tuple(x.data_ptr() for x in o)
for _ in range(30):
    rv.predict_read_device(*rp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    rv.predict_read_device(*rp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"device-resident read mode: {n} windows in {dt*1e3:.3f} ms = {n/dt/1e6:.2f} M bases/s ({n/4096:.2f} groups, {dt*1e3/np.ceil(n/4096):.4f} ms per group round)", flush=True)
for rep in range(4):
    t0 = time.perf_counter(); rv.predict_reads_raw(*raws); dt = time.perf_counter() - t0
    print(f"nrv_predict_reads_raw: {n/dt/1e6:.2f} M bases/s ({dt*1e3:.3f} ms)", flush=True)
