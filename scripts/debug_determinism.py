import sys
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn

desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
ctx = nh.NerfHip(0)
ctx.load_model(desc)
W, H = 1920, 1080
ctx.set_resolution(W, H)
cam, pose = syn.default_camera(W, H), syn.orbit_pose(30, 30)
frames = []
for i in range(6):
    ctx.render(cam, pose)
    a, d = ctx.read_f32()
    frames.append((a.copy(), d.copy(), ctx.stats().n_samples))
print("samples", [f[2] for f in frames])
ref = frames[0][0]
for i in range(1, 6):
    diff = np.abs(frames[i][0] - ref).max(axis=2)
    ys, xs = np.nonzero(diff)
    print(f"frame {i}: {len(ys)} px differ; max {diff.max():.3e}")
    for y, x in list(zip(ys, xs))[:12]:
        tile = (y // 8) * 240 + x // 8
        print(f"   px ({x},{y}) tile {tile} lane {(y%8)*8+x%8} block~{tile//4} d={diff[y,x]:.2e} a={frames[i][0][y,x]} ref={ref[y,x]}")
# network stage kernel determinism
rng = np.random.default_rng(0)
n = 1 << 20
xyz = torch.from_numpy(rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)).cuda()
dr = rng.normal(size=(n, 3)).astype(np.float32); dr /= np.linalg.norm(dr, axis=1, keepdims=True)
dr = torch.from_numpy(dr).cuda()
outs = []
for i in range(4):
    sig = torch.empty(n, device="cuda"); rgb = torch.empty((n, 3), device="cuda")
    torch.cuda.synchronize()
    ctx.network(xyz.data_ptr(), dr.data_ptr(), n, sig.data_ptr(), rgb.data_ptr())
    outs.append((sig.cpu().numpy(), rgb.cpu().numpy()))
for i in range(1, 4):
    print("network run", i, "sigma diffs", int((outs[i][0] != outs[0][0]).sum()), "rgb diffs", int((outs[i][1] != outs[0][1]).sum()))
# encode stage determinism
p01 = (xyz * 0.5 + 0.5).contiguous()
enc = []
for i in range(3):
    out = torch.empty((n, 32), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_grid(p01.data_ptr(), n, out.data_ptr())
    enc.append(out.cpu().numpy())
print("encode diffs", int((enc[1] != enc[0]).sum()), int((enc[2] != enc[0]).sum()))
