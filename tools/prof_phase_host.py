import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from cmbpo_amd import synthetic
B=1000; task="AntSafe-v2"
dev=torch.device("cuda:0")
w=bench.build_world(0,task)
sampler,pool,env,policy=bench.build_hip(w,task,B,dev,None,bench.MAXROLL,"schedule")
start=torch.from_numpy(synthetic.start_states(np.random.default_rng(1),B,task)).to(dev)
for _ in range(5): bench.rollout_phase(sampler,pool,start)
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
for _ in range(50): bench.rollout_phase(sampler,pool,start)
torch.cuda.synchronize(); pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
