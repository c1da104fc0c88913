"""Host-side profile of the train step (where does the Python/launch time go?)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model = bench.build_model(None, dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
pos, x, y = bench.make_inputs(32, 16384, 1234, dev)
for _ in range(3):
    bench.train_step(model, opt, pos, x, y)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    bench.train_step(model, opt, pos, x, y)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
