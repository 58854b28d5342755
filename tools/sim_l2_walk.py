#!/usr/bin/env python3
"""Host-side model of ONE XCD's L2 under different walk orders of the attention workgroups (round 4): 64 slots
run workgroups back to back, each walks a random 31.6 % list of 900 kv blocks at a speed drawn once per workgroup
(N(1, CV)) with 5 % step jitter; the L2 is an LRU of 64 K+V blocks.  Prints the hit rate.

    python tools/sim_l2_walk.py none|clock|jump CV [CHECK DELTA]

  none   ascending walk from entry 0 (the deterministic default)
  clock  start rotated to the phase of a clock cursor whose period is the nominal lifetime (JENGA_ATTN_ROTATE)
  jump   clock + every CHECK entries a workgroup that lags the cursor by more than 3*DELTA kv blocks moves the entries the
         cursor has already passed to the END of its walk and continues at the cursor (no waiting)

Model vs counters at CV = 0.10: none 8.5 % (measured 23.8 %: every head change re-aligns the real workgroups), clock 35 %
(measured 41.8 %), jump 55-60 % (built in round 4 and measured at 29 %: the model's pack runs at the cursor's speed by
construction, the real one does not -- DESIGN.md §3)."""
import collections
import heapq
import sys

import numpy as np


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "clock"
    cv = float(sys.argv[2]) if len(sys.argv) > 2 else 0.10
    check = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    delta = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    rng = np.random.default_rng(0)
    NB, SLOTS, CACHE, NWG, T = 900, 64, 64, 2712, 285.0
    lists = [np.sort(rng.choice(NB, size=int(rng.normal(285, 8)), replace=False)) for _ in range(NWG)]
    cache, events, state = collections.OrderedDict(), [], {}
    hits = miss = nxt = 0

    def start(slot, t):
        nonlocal nxt
        if nxt >= NWG:
            return
        L = lists[nxt]
        nxt += 1
        sp = max(0.6, rng.normal(1.0, cv))
        rot = int(((t % T) / T) * len(L)) if mode != "none" else 0
        state[slot] = [list(np.concatenate([L[rot:], L[:rot]])), 0, sp]
        heapq.heappush(events, (t + 1.0 / sp, slot))

    for s in range(SLOTS):
        start(s, 0.0)
    while events:
        t, slot = heapq.heappop(events)
        order, i, sp = state[slot]
        if mode == "jump" and i % check == 0 and i > 0 and len(order) - i > 2 * delta:
            ckv = ((t % T) / T) * NB
            lag = (ckv - order[i]) % NB
            if delta * 3 < lag < NB / 2:
                j = i
                while j < len(order) and 0 < ((ckv - order[j]) % NB) < NB / 2 and j - i < 60:
                    j += 1
                skipped = order[i:j]
                order[i:j] = []
                order.extend(skipped)
        b = int(order[i])
        if b in cache:
            hits += 1
            cache.move_to_end(b)
        else:
            miss += 1
            cache[b] = 1
            if len(cache) > CACHE:
                cache.popitem(last=False)
        i += 1
        if i >= len(order):
            start(slot, t)
        else:
            state[slot][1] = i
            heapq.heappush(events, (t + (1.0 / sp) * max(0.5, rng.normal(1.0, 0.05)), slot))
    print(f"{mode} cv={cv} hit_rate={hits / (hits + miss):.3f}")


if __name__ == "__main__":
    main()
