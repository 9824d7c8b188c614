import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vkit_amd import _native as N
ctx = N.default_ctx()
nb_up, nb_dn = 12582912, 13622715
hu = ctx.pinned_empty((nb_up,), np.uint8); hd = [ctx.pinned_empty((nb_dn,), np.uint8) for _ in range(4)]
du = [ctx.malloc(nb_up) for _ in range(4)]; dd = [ctx.malloc(nb_dn) for _ in range(4)]
def run(order, events, n=100):
    ctx.sync(); ctx.sync_stream(1); ctx.sync_stream(2)
    t0 = time.perf_counter(); evs = [None] * 4
    for k in range(n):
        s = k % 4
        if events and evs[s] is not None: ctx.event_wait(evs[s]); evs[s] = None
        ctx.upload_async(du[s], hu)
        if order: ctx.order(2, 0)
        ctx.download_async(dd[s], hd[s])
        if events: evs[s] = ctx.event_record(2)
    ctx.sync_stream(1); ctx.sync_stream(2)
    for e in evs:
        if e is not None: ctx.event_wait(e)
    return (time.perf_counter() - t0) / n * 1e3
for order in (0, 1):
    for events in (0, 1):
        print('order', order, 'events', events, 'ms per job', round(run(order, events), 3))
