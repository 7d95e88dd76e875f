"""Randomised equality run of the two schedulings of the render kernel (persistent against one workgroup per strip), poisoned
output planes, random frame sizes / shard layouts / view counts / cameras (outside, inside, far away, looking away):
usage: scripts/fuzz_persistent.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nerf-cuda_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import models, synthetic as syn
import test_persistent_gpu as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
built = [models.build_model(log2_hashmap_size=14, H=64), models.build_model(log2_hashmap_size=14, H=32, cascade=3, bound=4.0),
         models.build_model(log2_hashmap_size=13, H=32, dir_otype="Frequency", n_frequencies=12),                      # wide instance
         models.build_model(log2_hashmap_size=13, H=32, n_neurons=32, n_features_per_level=4, n_levels=8),             # generic, 12 waves
         models.build_model(log2_hashmap_size=13, H=64, cascade=2, bound=2.0, sh_degree=7, rgb_hidden_layers=3)]       # generic
descs = [b[0] for b in built]  # (`built` keeps the parameter arrays the descriptors point into alive)
bad = 0
for case in range(n_cases):
    W, H = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    count = int(rng.choice([1, 1, 2, 3, 5, 8]))
    index = int(rng.integers(0, count))
    n = int(rng.integers(1, 7)) if rng.random() < 0.95 else int(rng.integers(120, 140))  # sometimes more than one launch takes
    if n > 100:
        W, H = min(W, 96), min(H, 64)
    poses = []
    for _ in range(n):
        kind = rng.choice(["orbit", "inside", "away", "far"])
        poses.append(T._poses(kind, 3)[int(rng.integers(0, 3))])
    desc = descs[int(rng.integers(0, len(descs)))]
    try:
        ref = T._render(desc, W, H, poses, T.STRIP, shard=(index, count))
        got = T._render(desc, W, H, poses, T.PERSISTENT, shard=(index, count))
        same_px = np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32))
        if not same_px or got[2:] != ref[2:]:
            d = np.abs(got[0] - ref[0])
            print("DIFF", "model", descs.index(desc), W, H, index, count, n, "pixels equal", same_px, "max|d|", float(np.nanmax(d)), "differing px", int((d.max(axis=-1) > 0).sum()),
                  "samples/rays", got[2:], ref[2:], flush=True)
            if not same_px and bad < 3:
                for name, a, b in (("rgba", got[0], ref[0]), ("depth", got[1], ref[1])):
                    w = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
                    print("   ", name, "differing elements", len(w), "first", w[:4].tolist(), "values", [(float(a[tuple(i)]), float(b[tuple(i)])) for i in w[:4]], flush=True)
            bad += 1
    except AssertionError as e:
        bad += 1
        print("MISMATCH", W, H, index, count, n, str(e)[:200], flush=True)
    except Exception as e:  # an API error is a finding too
        bad += 1
        print("ERROR", W, H, index, count, n, str(e)[:300], flush=True)
print(f"{n_cases} random cases, {bad} mismatches", flush=True)
