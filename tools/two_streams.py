"""Experiment: one batch of n reads vs k concurrent sub-batches (one host thread + stream each)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mtsv_tools_amd as M

n = 1_000_000
ix = M.MGIndex.synth(0x6D747376, 256, 4, 270000, threads=32)
ix.to_device(0)
bases, off = M.synth_reads(ix, 1000, n, 150)
p = M.default_params()
for mode in (0, 1):
    for k in (1, 2, 3, 4):
        per = n // k
        bs = []
        for i in range(k):
            b = M.Batch(ix, 0, per, per * 150)
            b.set_verify_mode(mode)
            b.upload(bases[i * per * 150:(i + 1) * per * 150], off[i * per:(i + 1) * per + 1] - off[i * per])
            bs.append(b)
        def run_all():
            ts = [threading.Thread(target=b.run, args=(p,)) for b in bs]
            [t.start() for t in ts]; [t.join() for t in ts]
        run_all()
        t0 = time.perf_counter()
        for _ in range(5):
            run_all()
        dt = (time.perf_counter() - t0) / 5
        print(f"mode={mode} streams={k}: {dt*1e3:.2f} ms per {per*k} reads -> {per*k/dt/1e6:.1f} M reads/s", flush=True)
        for b in bs: b.close()
