import sys
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, oracle_py as op
desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
ctx = nh.NerfHip(0); ctx.load_model(desc)
o = op.Oracle(desc)
rng = np.random.default_rng(7)
n = 2_000_000
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
d01 = (0.5 * d + 0.5).astype(np.float32)
want = o.encode_dir(d01)
out = torch.empty((n, 16), dtype=torch.int16, device="cuda")
dd = torch.from_numpy(d01).cuda(); torch.cuda.synchronize()
ctx.encode_dir(dd.data_ptr(), n, out.data_ptr())
got = out.cpu().numpy().view(np.uint16)
bad = (got != want)
print("mismatching coefficients:", int(bad.sum()), "of", bad.size, "per column:", bad.sum(0).tolist())
