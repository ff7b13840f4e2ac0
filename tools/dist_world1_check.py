import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
import gpu_matrix_inversion_amd as g
dist.init_process_group("nccl", init_method="env://")
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
rng = np.random.default_rng(0)
a = torch.from_numpy(np.stack([(rng.uniform(-1, 1, (512, 512)) + 23 * np.eye(512)).astype(np.float32) for _ in range(3)])).cuda()
inv = g.Inverter()
out, st, worst, tm = g.invert_distributed(a, lambda s: inv.inv(s), root=0)
ref, _ = inv.inv(a)
print("world", dist.get_world_size(), "worst", worst, "identical", bool(torch.equal(out, ref)), {k: round(v * 1e3, 3) for k, v in tm.items()})
dist.destroy_process_group()
