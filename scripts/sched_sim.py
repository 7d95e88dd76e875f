"""List-scheduling model of the persistent kernel's tile queue on measured per-tile costs (scripts/tile_cost_dump.py):
what a cost-sorted queue order could give for ONE view.  4096 waves in 256 workgroups; a workgroup takes whole strips
(4 tiles) in queue order and deals them to its waves; once the queue is dry a workgroup's remaining work is shared by its
16 waves (tail splitting) at efficiency ETA."""
import heapq, sys
import numpy as np

ETA = 0.75

def simulate(strips, cost, split=True):
    """strips: list of strip ids in queue order; cost[strip] = 4 tile costs."""
    NW, WG = 4096, 256
    free = [(0.0, w) for w in range(NW)]
    heapq.heapify(free)
    wg_last = np.zeros(WG)       # per workgroup: per-wave end times
    ends = np.zeros(NW)
    t_dry = 0.0
    pending = {}                 # wg -> list of tiles of its current strip
    qi = 0
    # event-driven: the wave that becomes free first asks its workgroup's current strip, refilling from the queue
    cur = [[] for _ in range(WG)]
    while free:
        t, w = heapq.heappop(free)
        g = w // 16
        if not cur[g]:
            if qi < len(strips):
                cur[g] = list(cost[strips[qi]]); qi += 1
                if qi == len(strips): t_dry = t
            else:
                ends[w] = max(ends[w], t)
                continue
        c = cur[g].pop(0)
        ends[w] = t + c
        heapq.heappush(free, (t + c, w))
    e = ends.reshape(WG, 16)
    if not split:
        return e.max(), t_dry
    # tail splitting: from the moment a workgroup's first wave goes idle, the remaining work is shared
    out = 0.0
    for g in range(WG):
        ee = np.sort(e[g])
        # waves idle from ee[0]; iterate: at time t the idle waves help; model = remaining work after ee[0] spread over 16 waves / ETA
        t0 = ee[0]
        rem = (ee - t0).sum()
        out = max(out, t0 + rem / 16 / ETA)
    return out, t_dry

for az in (0, 45, 90, 135):
    p = np.load(f"gpurun_out/tile_cost_az{az}.npy"); start, cost = p[0], p[1]
    steps = np.load(f"gpurun_out/tile_steps_az{az}.npy")
    H, Wt = cost.shape  # 135 x 240
    sc = cost.reshape(H, Wt // 4, 4)
    n_strips = H * (Wt // 4)
    scost = sc.reshape(n_strips, 4)
    live = np.nonzero(scost.sum(1) > 0)[0]
    rows = live // (Wt // 4)
    # current order: centre-out rows of the ROI, columns left to right (classes ignored)
    r0, r1 = rows.min(), rows.max(); n = r1 - r0 + 1
    def co(i): off = (i + 1) >> 1; return (n - 1) // 2 + (off if i & 1 else -off)
    rank = {r0 + co(i): i for i in range(n)}
    cur_order = sorted(live, key=lambda s: (rank[s // (Wt // 4)], s % (Wt // 4)))
    true_lpt = sorted(live, key=lambda s: -scost[s].max())
    est = steps.reshape(n_strips, 4)
    est_lpt = sorted(live, key=lambda s: -est[s].max())
    est_sum = sorted(live, key=lambda s: -est[s].sum())
    def buckets(key, nb):
        k = np.array([key(s) for s in live]); edges = np.quantile(k[k > 0], np.linspace(0, 1, nb + 1)[1:-1]) if (k > 0).any() else []
        b = np.searchsorted(edges, k)
        return [s for _, _, s in sorted(zip(-b, [cur_order.index(s) for s in live], live))]
    total = scost[live].sum() / 4096
    print(f"az {az}: balanced {total:.3f}; measured end {(start + cost).max():.3f}")
    for name, order in (("current", cur_order), ("LPT true max", true_lpt), ("LPT steps max", est_lpt), ("LPT steps sum", est_sum),
                        ("8 buckets of steps max", buckets(lambda s: est[s].max(), 8)), ("4 buckets", buckets(lambda s: est[s].max(), 4))):
        a, td = simulate(order, scost, split=False)
        b, _ = simulate(order, scost, split=True)
        print(f"   {name:24s} no split {a:.3f}  split {b:.3f}  (queue dry at {td:.3f})")
