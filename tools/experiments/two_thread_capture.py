#!/usr/bin/env python3
"""Two extractor handles driven from two threads (src/Frame.cc:82-85), many rounds: every exception with its message."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from orbhip import capi, synth
imgs = [synth.synth_frame(30), synth.synth_frame(31)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fails = 0
for r in range(rounds):
    errs = [None, None]
    def work(i):
        try:
            ex = capi.Extractor()
            for _ in range(6):
                ex.extract(imgs[i])
        except BaseException as e:
            errs[i] = e
    stop = [False]
    def churn():                                   # a third thread creates, uses once and destroys handles meanwhile
        k = 0
        while not stop[0]:
            e = capi.Extractor(nfeatures=300 + 50 * (k % 3))
            e.extract(imgs[k & 1][:240, :320].copy())
            e.close() if hasattr(e, "close") else None
            del e
            k += 1
    tc = threading.Thread(target=churn) if os.environ.get("CHURN") else None
    if tc: tc.start()
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    stop[0] = True
    if tc: tc.join()
    if errs != [None, None]:
        fails += 1
        print("round", r, errs, flush=True)
print("rounds", rounds, "with an exception:", fails)
