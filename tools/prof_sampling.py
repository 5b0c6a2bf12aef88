import sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
os.environ["LGM_NO_SAMPLER_GRAPH"] = "1"
r = bench.run_sampling(dev, steps=12, batch=64, img=64)
print(r)
