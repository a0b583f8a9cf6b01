"""Lock-step simulation of the persistent kernel's ray scheduling (1024 waves x 64 lane slots, global queue, tail mode) on measured
per-ray sample counts (tests/tools/dump_counts.py writes gpurun_out/ray_counts.npz on a GPU box).  Reproduces the measured lane
utilisation of every variant and was used to decide on the longest-ray-first work list (DESIGN.md "Work-list order")."""
import numpy as np, heapq
c = np.load('gpurun_out/ray_counts.npz')['counts'].astype(np.int64)
N = c.size
# hit list order as k_first_hit produces: wave-compacted, block order roughly scanline (atomic order ~ arbitrary but near-monotone)
hit = np.nonzero(c > 0)[0]
def simulate(order, kmax=8, waves=1024):
    """event-free simulation: each wave holds 64 lanes; global queue; one 'round' = every active lane advances one sample
    (k>1 in tail mode: k samples per round for a ray). Waves advance in lock-step rounds (approximation: all rounds equal cost)."""
    q = list(order); qpos = 0
    lanes = np.zeros((waves, 64), np.int64)   # remaining samples per lane slot
    rounds = np.zeros(waves, np.int64)
    busy_lane_rounds = 0
    active = np.ones(waves, bool)
    total_rounds = 0
    # iterate global time in rounds; all waves step together (round cost constant)
    remaining = c[order].copy()
    t = 0
    drained_at = None
    while True:
        # refill
        need = (lanes == 0)
        n_need = int(need.sum())
        if qpos < len(q):
            take = min(n_need, len(q) - qpos)
            idx = np.argwhere(need)
            # fill in wave-major order but fair: interleave waves
            sel = idx[:take] if take == n_need else idx[np.random.default_rng(0).permutation(n_need)[:take]]
            lanes[sel[:, 0], sel[:, 1]] = c[q[qpos:qpos + take]]
            qpos += take
        act = lanes > 0
        per_wave = act.sum(1)
        if per_wave.sum() == 0:
            break
        drained = qpos >= len(q)
        if drained and drained_at is None: drained_at = t
        if drained:
            # tail mode: k = largest power of two <= kmax with per_wave*k <= 64
            k = np.ones(waves, np.int64)
            for kk in (2, 4, 8, 16, 32, 64):
                if kk <= kmax:
                    k = np.where((per_wave > 0) & (per_wave * kk <= 64), kk, k)
            dec = k[:, None] * act
        else:
            dec = act.astype(np.int64)
        busy_lane_rounds += int(np.minimum(lanes, dec).sum())
        lanes = np.maximum(lanes - dec, 0)
        rounds += (per_wave > 0)
        t += 1
    return t, rounds, busy_lane_rounds, drained_at
for name, order in [("scanline", hit), ("sorted desc (LPT)", hit[np.argsort(-c[hit], kind='stable')]), ("sorted asc", hit[np.argsort(c[hit], kind='stable')])]:
    for kmax in (1, 8):
        t, rounds, busy, dr = simulate(order, kmax)
        tot = rounds.sum()
        print(f"{name:20s} kmax {kmax}: makespan {t} rounds, sum of wave-rounds {tot}, ideal {c.sum()/ (1024*64):.1f}, lane util {c.sum()/(tot*64):.3f}, makespan eff {c.sum()/(t*1024*64):.3f}, drained at {dr}")

print("---- per-wave finish distribution")
def finish_dist(order, kmax):
    # re-run and record each wave's last active round
    q = list(order); qpos = 0; waves=1024
    lanes = np.zeros((waves, 64), np.int64); last = np.zeros(waves, np.int64); t = 0
    while True:
        need = (lanes == 0); n_need = int(need.sum())
        if qpos < len(q):
            take = min(n_need, len(q) - qpos); idx = np.argwhere(need)
            sel = idx[:take] if take == n_need else idx[np.random.default_rng(0).permutation(n_need)[:take]]
            lanes[sel[:, 0], sel[:, 1]] = c[q[qpos:qpos + take]]; qpos += take
        act = lanes > 0; per_wave = act.sum(1)
        if per_wave.sum() == 0: break
        if qpos >= len(q):
            k = np.ones(waves, np.int64)
            for kk in (2, 4, 8, 16, 32, 64):
                if kk <= kmax: k = np.where((per_wave > 0) & (per_wave * kk <= 64), kk, k)
            dec = k[:, None] * act
        else: dec = act.astype(np.int64)
        lanes = np.maximum(lanes - dec, 0); t += 1
        last[per_wave > 0] = t
    return last
lpt = hit[np.argsort(-(c[hit] >> 4), kind='stable')]
for name, order, kmax in [("scanline k8", hit, 8), ("LPT(16-wide buckets) k8", lpt, 8), ("LPT k64", lpt, 64)]:
    last = finish_dist(order, kmax)
    print(name, "finish round percentiles 0/10/50/90/100:", np.percentile(last, [0, 10, 50, 90, 100]).astype(int), "mean", last.mean().round(1))

print("---- integer k in tail mode")
def sim_intk(order, kmax, intk):
    q = list(order); qpos = 0; waves=1024
    lanes = np.zeros((waves, 64), np.int64); t = 0; rounds=np.zeros(waves,np.int64); last=np.zeros(waves,np.int64)
    while True:
        need = (lanes == 0); n_need = int(need.sum())
        if qpos < len(q):
            take = min(n_need, len(q) - qpos); idx = np.argwhere(need)
            sel = idx[:take] if take == n_need else idx[np.random.default_rng(0).permutation(n_need)[:take]]
            lanes[sel[:, 0], sel[:, 1]] = c[q[qpos:qpos + take]]; qpos += take
        act = lanes > 0; per_wave = act.sum(1)
        if per_wave.sum() == 0: break
        if qpos >= len(q):
            if intk:
                k = np.where(per_wave > 0, np.minimum(kmax, 64 // np.maximum(per_wave, 1)), 1)
            else:
                k = np.ones(waves, np.int64)
                for kk in (2, 4, 8, 16, 32, 64):
                    if kk <= kmax: k = np.where((per_wave > 0) & (per_wave * kk <= 64), kk, k)
            dec = k[:, None] * act
        else: dec = act.astype(np.int64)
        lanes = np.maximum(lanes - dec, 0); t += 1
        rounds += per_wave > 0; last[per_wave > 0] = t
    return t, rounds.sum(), last.mean()
for kmax, intk in [(8, False), (8, True), (16, True), (64, True)]:
    t, tot, mean = sim_intk(lpt, kmax, intk)
    print(f"LPT kmax {kmax} integer-k {intk}: makespan {t}, wave-rounds {tot}, mean finish {mean:.1f}, util {c.sum()/(tot*64):.3f}")

print("---- bucket shapes (tail mode kmax 8)")
def key_var(cost, thresh, wide, fine):
    return np.where(cost >= thresh, (thresh >> fine) + ((cost - thresh) >> wide) + 1, cost >> fine)
print("cost histogram (hit rays):", np.histogram(c[hit], bins=[1, 8, 16, 24, 32, 40, 48, 56, 64, 80])[0])
for name, key in [("uniform 16", c[hit] >> 4), ("uniform 4", c[hit] >> 2), ("uniform 1", c[hit]),
                  ("16 above 32, 2 below", key_var(c[hit], 32, 4, 1)), ("16 above 48, 4 below", key_var(c[hit], 48, 4, 2)),
                  ("16 above 16, 1 below", key_var(c[hit], 16, 4, 0))]:
    order = hit[np.argsort(-key, kind='stable')]
    t, tot, mean = sim_intk(order, 8, False)
    print(f"{name:24s}: makespan {t}, mean finish {mean:.1f}, util {c.sum()/(tot*64):.3f}, distinct buckets {len(np.unique(key))}")


print("---- k lanes per ray from the start (k0), doubling in tail mode up to kmax")
def sim_k0(order, k0, kmax=8, waves=1024):
    q = list(order); qpos = 0
    slots = 64 // k0
    rem = np.zeros((waves, slots), np.int64); t = 0; rounds = 0; last = np.zeros(waves, np.int64)
    while True:
        need = rem == 0; n_need = int(need.sum())
        if qpos < len(q):
            take = min(n_need, len(q) - qpos); idx = np.argwhere(need)
            sel = idx[:take] if take == n_need else idx[np.random.default_rng(0).permutation(n_need)[:take]]
            rem[sel[:, 0], sel[:, 1]] = c[q[qpos:qpos + take]]; qpos += take
        act = rem > 0; r = act.sum(1)
        if r.sum() == 0: break
        k = np.full(waves, k0, np.int64)
        if qpos >= len(q):
            for kk in (2, 4, 8, 16, 32, 64):
                if k0 < kk <= kmax: k = np.where((r > 0) & (r * kk <= 64), kk, k)
        rem = np.maximum(rem - k[:, None] * act, 0); t += 1
        rounds += int((r > 0).sum()); last[r > 0] = t
    return t, rounds, last.mean()
for k0 in (1, 2, 4):
    t, tot, mean = sim_k0(lpt, k0)
    print(f"k0 {k0}: makespan {t} rounds, wave-rounds {tot}, mean finish {mean:.1f}  (ideal {c.sum()/65536:.1f})")
